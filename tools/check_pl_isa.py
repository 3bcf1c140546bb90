"""ISA check of the plane kernels' main loops (csrc/gemm_pl.hip, csrc/gemm_plw.hip), in the spirit of tools/check_sk_isa.py: the K loop of
every instantiation must hold 48 MFMAs, its fragment reads (24 ds_read_b128, or 48 ds_read_b64_tr_b16 in the weight-gradient kernel),
9 LDS-DMA instructions, exactly one s_barrier, NO scratch access and no compiler-inserted `s_waitcnt vmcnt`
besides the loop's own.  Usage: python tools/check_pl_isa.py [pl|plw] [path/to/kernel.s]  (compiles the source when no path is given).
Prints one line per instantiation (`lanes` = v_readlane / v_writelane of SGPR spills inside the loop: reported, not judged), exit code 1 on a
violation."""
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "comprehensive-transformer-tts_amd", "csrc")
KERNELS = {"pl": ("gemm_pl.hip", "gemm_pl_kernel", "ds_read_b128", 24), "plw": ("gemm_plw.hip", "gemm_plw_kernel", "ds_read_b64_tr_b16", 48)}


def compile_isa(which="pl"):
    SRC = os.path.join(CSRC, KERNELS[which][0])
    out = os.path.join(tempfile.mkdtemp(prefix="plisa"), "kernel.s")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only", SRC, "-o", out],
                   check=True, stderr=subprocess.DEVNULL)
    return out


def loops(body):
    """(start, end) line ranges of the innermost backward branches"""
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    res = []
    for i, l in enumerate(body):
        m = re.search(r"s_cbranch\w*\s+(\.LBB\d+_\d+)", l) or re.search(r"s_branch\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            res.append((labels[m.group(1)], i))
    return res


def check(path, which="pl"):
    _, kname, rd, n_rd = KERNELS[which]
    lines = open(path).read().split("\n")
    ok = True
    # the six-term instantiations (template argument TERMS = 6: `...ILb?ELi6EE...`); the one-term "amp" variants are not judged here
    starts = [i for i, l in enumerate(lines) if re.match(r"^_ZN.*" + kname + r"ILb\dELi6E.*:", l)]
    for st in starts:
        name = lines[st].split(":")[0]
        end = next(i for i in range(st, len(lines)) if ".Lfunc_end" in lines[i])
        body = lines[st:end]
        mf = [i for i, l in enumerate(body) if "v_mfma" in l]
        # the K loop exists once per wave group (two instruction orders): runs of 48 MFMAs
        groups = [mf[k:k + 48] for k in range(0, len(mf), 48)]
        for gi, grp in enumerate(groups):
            # the K loop proper: from the loop head (the nearest backward-branch target in front of the group) to the first branch behind
            # the block's last MFMA / last LDS-DMA instruction, whichever comes later (the `continue` of `if (ckb < cp.kb_hi) continue;`)
            a = max(a for a, b in loops(body) if a <= grp[0] and b >= grp[-1])
            dmas = [i for i in range(a, len(body)) if "buffer_load_dwordx4" in body[i] and " lds" in body[i]][:9]
            last = max([grp[-1]] + dmas)
            cont = next(i for i in range(last, len(body)) if re.search(r"s_cbranch|s_branch", body[i]))
            seg = body[a:cont + 1]
            cnt = lambda pat: sum(1 for l in seg if re.match(r"\s+" + pat, l))
            n = dict(mfma=cnt("v_mfma"), frag_reads=cnt(rd), lanes=cnt("v_readlane") + cnt("v_writelane"), dma=sum(1 for l in seg if "buffer_load_dwordx4" in l and " lds" in l),
                     barrier=cnt("s_barrier"), scratch=cnt("scratch_"), vm_wait=sum(1 for l in seg if re.match(r"\s+s_waitcnt.*vmcnt", l)),
                     instr=sum(1 for l in seg if re.match(r"\s+[sv]_|\s+ds_|\s+buffer_|\s+global_|\s+scratch_", l)))
            good = n["mfma"] == 48 and n["frag_reads"] == n_rd and n["dma"] == 9 and n["barrier"] == 1 and n["scratch"] == 0 and n["vm_wait"] <= 1
            ok &= good
            print(("ok  " if good else "BAD ") + name[-40:] + f" loop {gi}", n)
    return ok


if __name__ == "__main__":
    args = sys.argv[1:]
    which = args.pop(0) if args and args[0] in KERNELS else "pl"
    sys.exit(0 if check(args[0] if args else compile_isa(which), which) else 1)

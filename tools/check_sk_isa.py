"""Static check of the stream-K GEMM's K loop (csrc/gemm_sk.hip): between the loop's own `s_waitcnt vmcnt(0); s_barrier` and the
end of the MFMA chain hipcc must not have inserted any vmcnt wait of its own - such a wait drains the DMA of the NEXT K-block and
serialises load and compute.  Compiles the file to ISA (no GPU needed) and inspects every gemm_sk_kernel instantiation.
usage: python tools/check_sk_isa.py [extra hipcc flags]   -> one line per kernel, exit code 1 on a violation"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "comprehensive-transformer-tts_amd", "csrc", "gemm_sk.hip")


def kernels_isa(extra=()):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "sk.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "--cuda-device-only", "-S", SRC, "-o", out,
                        *extra], check=True, capture_output=True)
        text = open(out).read()
    ks = {}
    for m in re.finditer(r"^(_ZN\S*gemm_sk_kernel[^\s:]*):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M):
        ks[m.group(1)] = m.group(2).split("\n")
    return ks


def check(lines):
    """returns (n_loop_mfma, violations): the loop = from the asm `s_barrier` to the last v_mfma before the next scalar back-branch"""
    bar = [i for i, l in enumerate(lines) if l.strip() == "s_barrier" and "ASMSTART" in "".join(lines[max(0, i - 3):i])]
    if not bar:
        return 0, ["no loop barrier found"]
    start = bar[0]
    mf = [i for i, l in enumerate(lines) if "v_mfma" in l and i > start]
    # the MFMA chain of the loop = the first contiguous group of 16 after the barrier
    # the MFMA chain of the loop = the first group of consecutive v_mfma lines (at most 24 other lines between two of them)
    chain = mf[:1]
    for i in mf[1:]:
        if i - chain[-1] > 24:
            break
        chain.append(i)
    if len(chain) < 16 or len(chain) % 16:
        return len(chain), [f"{len(chain)} MFMAs after the loop barrier"]
    in_asm, bad = False, []
    for i in range(start + 1, chain[-1]):
        if "ASMSTART" in lines[i]:
            in_asm = True
        elif "ASMEND" in lines[i]:
            in_asm = False
        elif "vmcnt" in lines[i] and not in_asm:          # the kernel's own waits are inline asm; anything else is hipcc's
            # a wait right behind a register-spill reload sits in the piece-change path (executed once per tile piece, not per K-block):
            # it drains one prefetch there - tolerated, everything else is a violation
            if any("scratch_load" in lines[j] for j in range(max(0, i - 32), i)):
                continue
            bad.append(f"line {i}: {lines[i].strip()}")
    return len(chain), bad


def main():
    rc = 0
    for name, lines in sorted(kernels_isa(sys.argv[1:]).items()):
        n, bad = check(lines)
        tag = re.search(r"ILb(\d)ELb(\d)ELb(\d)ELi(\d)ELi(\d)ELi(\d)E", name)
        print(f"gemm_sk_kernel<A_KC={tag.group(1)},B_KC={tag.group(2)},CONV={tag.group(3)},STAGES={tag.group(4)},MT={tag.group(5)},NT={tag.group(6)}>: "
              f"{n} MFMAs, {'ok' if not bad else 'VIOLATION ' + '; '.join(bad)}")
        rc |= bool(bad)
    return rc


if __name__ == "__main__":
    sys.exit(main())

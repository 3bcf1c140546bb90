"""hipcc must not put a vmcnt wait between the LDS-DMA of the next half and the MFMAs of the current one (gemm_ws.hip): list every
s_waitcnt with a vmcnt term inside the MFMA span of each gemm_ws_kernel instantiation that is not one of the two explicit syncs.
python tools/check_ws_isa.py [asm file]  (default: compile csrc/gemm_ws.hip to /tmp)"""
import os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def compile_asm(out="/tmp/gemm_ws_check.s"):
    src = os.path.join(ROOT, "comprehensive-transformer-tts_amd", "csrc", "gemm_ws.hip")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", src, "-o", out],
                   check=True, stderr=subprocess.DEVNULL)
    return out


def check(path):
    lines = open(path).read().split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"_ZN.*gemm_ws_kernel.*:\s", l)]
    report = []
    for k, s in enumerate(starts):
        e = next(i for i in range(s, len(lines)) if "s_endpgm" in lines[i])
        L = lines[s:e]
        mf = [i for i, l in enumerate(L) if "v_mfma" in l]
        # a compiler wait hurts between the DMA of the next half and the end of the MFMA chain that follows it (the epilogue behind the
        # chain may wait for its own loads: the DMA has had a whole half to land by then)
        dma = [i for i, l in enumerate(L) if "buffer_load" in l and " lds" in l]
        bad = []
        for k, i in enumerate(dma):
            if k + 1 < len(dma) and dma[k + 1] - i < 40:
                continue                                  # not the last DMA instruction of its group
            chain = [j for j in mf if j > i]
            if not chain or chain[0] - i > 400:
                continue
            end = chain[0]
            for j in chain[1:]:
                if j - end > 30:
                    break
                end = j
            bad += [L[j].strip() for j in range(i, end) if "vmcnt" in L[j] and "s_barrier" not in L[j + 1]
                    and not any("s_barrier" in x for x in L[max(0, j - 4):j])]
        spill = any("scratch_" in l for l in L)
        report.append(dict(name=lines[s].split(":")[0], n_mfma=len(mf), stray_vmcnt=bad, scratch=spill))
    return report


if __name__ == "__main__":
    rep = check(sys.argv[1] if len(sys.argv) > 1 else compile_asm())
    for r in rep:
        print(r["name"][-40:], r["n_mfma"], "stray:", r["stray_vmcnt"][:2], "scratch" if r["scratch"] else "")
    print(len(rep), "kernels;", sum(bool(r["stray_vmcnt"]) for r in rep), "with stray vmcnt waits")

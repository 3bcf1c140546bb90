"""CPU (hipcc cross-compiles without a GPU): no gemm_ws_kernel instantiation may contain a compiler-inserted `s_waitcnt vmcnt` between
the LDS-DMA of the next A half and the MFMAs of the current one, nor scratch (the 128 weight registers + accumulators + fragments sit
at 210-246 of 256 VGPRs).  Such a wait drains the DMA in front of the fragment reads and serialises load and compute; it appeared
three times while the kernel was written (weight loads first used inside the loop; epilogue loads whose results are unused on some
path; a debug flag that raised the register pressure).  tools/check_ws_isa.py compiles csrc/gemm_ws.hip to ISA and inspects it."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_weight_stationary_kernels_have_no_stray_vmcnt_waits_and_no_scratch(tmp_path):
    import check_ws_isa
    rep = check_ws_isa.check(check_ws_isa.compile_asm(str(tmp_path / "gemm_ws.s")))
    assert len(rep) == 48, len(rep)                      # 2 layouts x 4 activations x (fwd: drop x residual, bwd: drop)
    for r in rep:
        assert r["n_mfma"] % 256 == 0 and r["n_mfma"] > 0, r
        assert not r["stray_vmcnt"], r
        assert not r["scratch"], r

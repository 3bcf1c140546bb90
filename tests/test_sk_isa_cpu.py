"""CPU (hipcc cross-compiles without a GPU): the K loop of every gemm_sk_kernel instantiation must be free of compiler-inserted
`s_waitcnt vmcnt` - such a wait drains the LDS-DMA of the next K-block and silently serialises load and compute (it happened twice
while the kernel was written: in front of a fragment read whose registers an epilogue load had used, and behind a debug store on the
loop's exit edge).  tools/check_sk_isa.py compiles csrc/gemm_sk.hip to ISA and inspects the loop of all instantiations."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stream_k_loops_have_no_compiler_inserted_vmcnt_waits():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_sk_isa.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("gemm_sk_kernel")]
    assert len(lines) >= 18 and all(l.endswith("ok") for l in lines), r.stdout

#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd $ROOT; timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "stream_k or conv or persistent" 2>&1 | tail -3; cd /tmp
for o in 0 1; do echo "== CONV_ORDER=$o"; CTTS_SK_CONV_ORDER=$o timeout 300 python $ROOT/tools/bench_sk.py 60 "ffn1" 2>&1 | grep -v amdgpu.ids | cut -c1-150 | head -4; CTTS_SK_CONV_ORDER=$o timeout 300 python $ROOT/tools/bench_sk.py 60 "postnet conv fwd" 2>&1 | grep -v amdgpu.ids | cut -c1-150; done
for o in 0 1 0 1; do echo "fs2 CONV_ORDER=$o"; CTTS_SK_CONV_ORDER=$o timeout 300 python $ROOT/bench.py --no-cpu-baseline --no-pcie --no-secondary --no-roofline 2>/dev/null | tail -1 | cut -c80-200; done

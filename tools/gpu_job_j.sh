#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp; cd /tmp
for sk in 0 1 0 1; do
  echo "== bench.py fs2 CTTS_SK=$sk"; CTTS_SK=$sk timeout 300 python $ROOT/bench.py --no-cpu-baseline --no-pcie 2>/dev/null | tail -1 | cut -c80-220
done
for sk in 0 1; do
  echo "== bench.py conformer CTTS_SK=$sk"; CTTS_SK=$sk timeout 300 python $ROOT/bench.py --block conformer --steps 10 --warmup 3 --no-cpu-baseline --no-pcie 2>/dev/null | tail -1 | cut -c80-220
done
CTTS_SK=1 timeout 200 python /root/repo/tools/profile_gemm_shapes.py 2>&1 | grep -v amdgpu.ids | head -12

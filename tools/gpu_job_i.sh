#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp; cd /tmp
echo "== default (128x128, K>=2048)"; timeout 200 python $ROOT/tools/bench_sk.py 20 2>&1 | grep -v amdgpu.ids | grep -v "NT \|NN\|TN\|epilogue\|mel" | cut -c1-140
for sk in 0 1 0 1; do
  echo "== bench.py fs2 CTTS_SK=$sk"; CTTS_SK=$sk timeout 300 python $ROOT/bench.py --no-cpu-baseline --no-pcie 2>/dev/null | tail -1 | cut -c80-220
done
for sk in 0 1; do
  echo "== bench.py conformer CTTS_SK=$sk"; CTTS_SK=$sk timeout 300 python $ROOT/bench.py --block conformer --steps 10 --warmup 3 --no-cpu-baseline --no-pcie 2>/dev/null | tail -1 | cut -c80-220
done

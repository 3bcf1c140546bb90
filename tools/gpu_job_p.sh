#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd $ROOT; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4; cd /tmp
timeout 300 python $ROOT/tools/bench_ws.py 100 N=1024 2>&1 | grep -v amdgpu.ids | cut -c1-150
for i in 1 2; do timeout 300 python $ROOT/bench.py --block conformer --steps 10 --warmup 3 --no-cpu-baseline --no-pcie --no-roofline 2>/dev/null | tail -1 | cut -c80-200; done

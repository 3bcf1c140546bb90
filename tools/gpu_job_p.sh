#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd $ROOT; timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q 2>&1 | tail -3; cd /tmp
for i in 1 2; do timeout 300 python $ROOT/bench.py --no-cpu-baseline --no-pcie --no-secondary --no-roofline 2>/dev/null | tail -1 | cut -c80-200; done
for i in 1 2; do timeout 300 python $ROOT/bench.py --block conformer --steps 10 --warmup 3 --no-cpu-baseline --no-pcie --no-roofline 2>/dev/null | tail -1 | cut -c80-200; done

#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd $ROOT
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_kernels_gpu.py tests/test_dp_gpu.py -m gpu -x -q -k "conformer or fusion or dwconv or batch_norm or bn or dp" 2>&1 | tail -8
cd /tmp
for i in 1 2; do timeout 300 python $ROOT/bench.py --block conformer --steps 10 --warmup 3 --no-cpu-baseline --no-pcie 2>/dev/null | tail -1 | cut -c80-200; done
timeout 300 python $ROOT/bench.py --no-cpu-baseline --no-pcie --no-secondary 2>/dev/null | tail -1 | cut -c80-200
python $ROOT/tools/find_torch_ops.py --block conformer 2>&1 | grep -v amdgpu | head -14

#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd /tmp
mkdir -p $ROOT/gpurun_out
timeout 300 python $ROOT/tools/bench_sk.py 60 ffn1 2>&1 | grep -v amdgpu.ids | cut -c1-150
cd $ROOT; timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q 2>&1 | tail -3; cd /tmp
for i in 1 2; do timeout 300 python $ROOT/bench.py --no-cpu-baseline --no-pcie --no-secondary --no-roofline 2>/dev/null | tail -1 | cut -c80-200; done
for i in 1 2; do timeout 300 python $ROOT/bench.py --block conformer --steps 10 --warmup 3 --no-cpu-baseline --no-pcie --no-roofline 2>/dev/null | tail -1 | cut -c80-200; done
timeout 300 python $ROOT/tools/profile_gemm_shapes.py > $ROOT/gpurun_out/shapes_fs2.txt 2>&1

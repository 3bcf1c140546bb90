#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd /tmp
for at in 0 1; do echo "== SK_ATOMIC=$at"; CTTS_SK_ATOMIC=$at timeout 300 python $ROOT/tools/bench_sk.py 100 "wgrad" 2>&1 | grep -v amdgpu.ids | cut -c1-150; done
for at in 0 1 0 1; do echo "fs2 SK_ATOMIC=$at"; CTTS_SK_ATOMIC=$at timeout 300 python $ROOT/bench.py --no-cpu-baseline --no-pcie --no-secondary --no-roofline 2>/dev/null | tail -1 | cut -c80-200; done
for at in 0 1 0 1; do echo "conformer SK_ATOMIC=$at"; CTTS_SK_ATOMIC=$at timeout 300 python $ROOT/bench.py --block conformer --steps 10 --warmup 3 --no-cpu-baseline --no-pcie --no-roofline 2>/dev/null | tail -1 | cut -c80-200; done

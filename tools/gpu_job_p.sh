#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for cfg in "fs2:" "conformer:--block conformer"; do
  n=${cfg%%:*}; a=${cfg#*:}; rm -rf /tmp/prof_$n
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$n -- python $ROOT/bench.py $a --no-graph --no-cpu-baseline --no-pcie --no-secondary --no-roofline --steps 5 --warmup 2 > /tmp/prof_$n.log 2>&1
  python $ROOT/tools/rocpd_summary.py $(find /tmp/prof_$n -name "*results.db" | head -1) 60 > $OUT/now_${n}_eager_kernel_stats.md 2>&1
done

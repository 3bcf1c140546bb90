#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd /tmp
for r in 0 4096 0 4096 100000; do echo "fs2 side_rows=$r"; CTTS_WGRAD_SIDE_ROWS=$r timeout 300 python $ROOT/bench.py --no-cpu-baseline --no-pcie --no-secondary --no-roofline 2>/dev/null | tail -1 | cut -c80-200; done
for r in 0 4096 0 4096; do echo "conformer side_rows=$r"; CTTS_WGRAD_SIDE_ROWS=$r timeout 300 python $ROOT/bench.py --block conformer --steps 10 --warmup 3 --no-cpu-baseline --no-pcie --no-roofline 2>/dev/null | tail -1 | cut -c80-200; done

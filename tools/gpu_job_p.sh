#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd /tmp
for f in 0 1 0 1; do echo "fs2 ws=$f"; CTTS_WS=$f timeout 300 python $ROOT/bench.py --no-cpu-baseline --no-pcie --no-secondary --no-roofline 2>/dev/null | tail -1 | cut -c80-200; done
for f in 0 1 0 1; do echo "conformer ws=$f"; CTTS_WS=$f timeout 300 python $ROOT/bench.py --block conformer --steps 10 --warmup 3 --no-cpu-baseline --no-pcie --no-roofline 2>/dev/null | tail -1 | cut -c80-200; done

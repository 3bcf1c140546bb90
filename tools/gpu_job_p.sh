#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd $ROOT; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4; cd /tmp
for i in 1 2; do timeout 300 python $ROOT/bench.py --no-cpu-baseline --no-pcie --no-secondary --no-roofline 2>/dev/null | tail -1 | cut -c80-200; done

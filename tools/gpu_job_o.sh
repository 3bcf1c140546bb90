#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
cd /tmp
for f in 0 1 0 1; do echo "fuse_epi=$f"; CTTS_FUSE_EPI_BWD=$f timeout 300 python $ROOT/bench.py --no-cpu-baseline --no-pcie --no-secondary 2>/dev/null | tail -1 | cut -c80-200; done
for f in 0 1; do echo "conformer fuse_epi=$f"; CTTS_FUSE_EPI_BWD=$f timeout 300 python $ROOT/bench.py --block conformer --steps 10 --warmup 3 --no-cpu-baseline --no-pcie 2>/dev/null | tail -1 | cut -c80-200; done

#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd /tmp
for o in 0 1; do
  for w in ffn1_step dgrad; do
  rm -rf /tmp/pmc; CTTS_SK_CONV_ORDER=$o timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc -- python $ROOT/tools/bench_one.py $w 30 > /tmp/pmc.log 2>&1
  echo "## order=$o $w FETCH_SIZE KiB (x2 on gfx950)"; python $ROOT/tools/rocpd_pmc_summary.py $(find /tmp/pmc -name "*results.db" | head -1) 2>&1 | grep -E "gemm_" | cut -c1-200
  done
done

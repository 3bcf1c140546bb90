#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
L=$ROOT/comprehensive-transformer-tts_amd/csrc
run() { echo "== $1"; shift; env "$@" timeout 100 python $ROOT/tools/bench_sk.py 20 "ffn1 fwd" 2>&1 | grep -v amdgpu.ids; }
run "baseline S2"  X=1
run "no DMA (debug 1)" CTTS_SK_DEBUG=1
run "same tile (debug 2)" CTTS_SK_DEBUG=2
run "S3 W96" CTTS_SK_STAGES=3
run "S3 W96 reads-first" CTTS_SK_STAGES=3 CTTS_LIB=$L/libctts_hip_rf.so
run "S3 W64" CTTS_SK_STAGES=3 CTTS_SK_W=64
run "S2 W64" CTTS_SK_W=64
run "S2 W32 (1 wg/CU)" CTTS_SK_W=32
run "S3 W32 (1 wg/CU)" CTTS_SK_STAGES=3 CTTS_SK_W=32
run "no DMA W32" CTTS_SK_DEBUG=1 CTTS_SK_W=32
run "no DMA W64" CTTS_SK_DEBUG=1 CTTS_SK_W=64
: > $OUT/r03_pmc_traffic_sk.md
for P in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  rm -rf /tmp/pmc; timeout 200 rocprofv3 --kernel-trace --pmc $P -d /tmp/pmc -- python $ROOT/tools/bench_sk.py 8 "ffn1 fwd" > /tmp/pmc.log 2>&1
  python $ROOT/tools/rocpd_pmc_summary.py $(find /tmp/pmc -name "*results.db" | head -1) 2>&1 < /dev/null | grep -E "^\| kernel|gemm_" | cut -c1-300 >> $OUT/r03_pmc_traffic_sk.md
done
cat $OUT/r03_pmc_traffic_sk.md

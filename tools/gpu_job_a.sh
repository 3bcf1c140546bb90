#!/bin/bash
# round-3 experiment batch A: stream-K variants, in-step A/B, PMC of the SK kernel
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
L=$ROOT/comprehensive-transformer-tts_amd/csrc
echo "== bench_one old path"; CTTS_SK=0 timeout 100 python $ROOT/tools/bench_one.py ffn1_step 30 2>&1 | tail -1
echo "== bench_sk default W=128"; timeout 200 python $ROOT/tools/bench_sk.py 20 2>&1 | grep -v amdgpu.ids
echo "== reads-first"; CTTS_LIB=$L/libctts_hip_rf.so timeout 100 python $ROOT/tools/bench_sk.py 20 ffn1 2>&1 | grep -v amdgpu.ids
echo "== W=160"; CTTS_SK_W=160 timeout 100 python $ROOT/tools/bench_sk.py 20 ffn1 2>&1 | grep -v amdgpu.ids
echo "== W=160 reads-first"; CTTS_SK_W=160 CTTS_LIB=$L/libctts_hip_rf.so timeout 100 python $ROOT/tools/bench_sk.py 20 ffn1 2>&1 | grep -v amdgpu.ids
echo "== W=96"; CTTS_SK_W=96 timeout 100 python $ROOT/tools/bench_sk.py 20 ffn1 2>&1 | grep -v amdgpu.ids
echo "== gw=16 (one group)"; CTTS_SK_GW=16 timeout 100 python $ROOT/tools/bench_sk.py 20 "ffn1 fwd" 2>&1 | grep -v amdgpu.ids
echo "== gw=2"; CTTS_SK_GW=2 timeout 100 python $ROOT/tools/bench_sk.py 20 "ffn1 fwd" 2>&1 | grep -v amdgpu.ids
for sk in 0 1; do
  echo "== bench.py fs2 CTTS_SK=$sk"; CTTS_SK=$sk timeout 300 python $ROOT/bench.py --no-cpu-baseline --no-pcie 2>/dev/null | tail -1 | cut -c1-400
done
for sk in 0 1; do
  echo "== bench.py conformer CTTS_SK=$sk"; CTTS_SK=$sk timeout 300 python $ROOT/bench.py --block conformer --steps 10 --warmup 3 --no-cpu-baseline --no-pcie 2>/dev/null | tail -1 | cut -c1-400
done
P1="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"
P2="SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS"
: > $OUT/r03_pmc_sq_gemm_sk.md
for P in "$P1" "$P2"; do
  rm -rf /tmp/pmc; timeout 200 rocprofv3 --kernel-trace --pmc $P -d /tmp/pmc -- python $ROOT/tools/bench_sk.py 8 "ffn1 fwd" > /tmp/pmc.log 2>&1
  python $ROOT/tools/rocpd_pmc_summary.py $(find /tmp/pmc -name "*results.db" | head -1) 2>&1 < /dev/null | grep -E "^\| kernel|gemm_" | cut -c1-400 >> $OUT/r03_pmc_sq_gemm_sk.md
done
cat $OUT/r03_pmc_sq_gemm_sk.md

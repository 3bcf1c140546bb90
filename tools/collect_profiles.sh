#!/bin/bash
# Round artefacts in one GPU-box call:  bash tools/collect_profiles.sh r02   -> gpurun_out/<round>_*  (copy the ones to keep into profiles/)
# bench lines of every BASELINE configuration, rocprofv3 kernel-trace summaries (eager launches), PMC passes (each in its own run:
# FETCH_SIZE and WRITE_SIZE do not fit one pass), micro-benchmarks, and the 2-rank data-parallel path on one GPU (gloo stand-in).
R=${1:-r02}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
b() { timeout 300 python $ROOT/bench.py "$@" 2>>$OUT/${R}_bench.err; }
b --steps 20 --warmup 5 > $OUT/${R}_bench_fs2.json
b --block conformer --steps 10 --warmup 3 --no-cpu-baseline > $OUT/${R}_bench_conformer.json
b --learn-alignment --steps 10 --warmup 3 --no-cpu-baseline --no-pcie > $OUT/${R}_bench_unsup.json
b --prosody liu2021 --steps 10 --warmup 3 --no-cpu-baseline --no-pcie > $OUT/${R}_bench_liu2021.json
b --prosody liu2021 --learn-alignment --steps 10 --warmup 3 --no-cpu-baseline --no-pcie > $OUT/${R}_bench_c5.json
b --dataset VCTK --steps 10 --warmup 3 --no-cpu-baseline --no-pcie > $OUT/${R}_bench_vctk.json
CTTS_FUSED_ATTN=1 b --steps 20 --warmup 5 --no-cpu-baseline --no-pcie > $OUT/${R}_bench_fs2_fused_attention.json
CTTS_FUSED_ATTN=0 b --block conformer --steps 10 --warmup 3 --no-cpu-baseline --no-pcie > $OUT/${R}_bench_conformer_unfused_attention.json
for s in weak strong; do
  CTTS_BENCH_SAME_DEVICE=1 CTTS_BENCH_BACKEND=gloo b --gpus 2 --scaling $s --steps 5 --warmup 2 --no-cpu-baseline --no-pcie > $OUT/${R}_bench_dp2_one_gpu_gloo_$s.json
done
for cfg in "fs2:" "conformer:--block conformer"; do
  n=${cfg%%:*}; a=${cfg#*:}; rm -rf /tmp/prof_$n
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$n -- python $ROOT/bench.py $a --no-graph --no-cpu-baseline --no-pcie --steps 5 --warmup 2 > /tmp/prof_$n.log 2>&1
  python $ROOT/tools/rocpd_summary.py $(find /tmp/prof_$n -name "*results.db" | head -1) 45 > $OUT/${R}_${n}_eager_kernel_stats.md 2>&1
done
rm -rf /tmp/prof_stft; timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_stft -- python $ROOT/tools/bench_stft.py > $OUT/${R}_bench_stft.jsonl 2>/dev/null
python $ROOT/tools/rocpd_summary.py $(find /tmp/prof_stft -name "*results.db" | head -1) 12 > $OUT/${R}_stft_kernel_stats.md 2>&1
timeout 100 python $ROOT/tools/bench_stft.py > $OUT/${R}_bench_stft.jsonl 2>/dev/null
timeout 200 python $ROOT/tools/bench_attn.py fs2 conformer > $OUT/${R}_microbench_attention.txt 2>&1
: > $OUT/${R}_pmc_traffic_gemm_shapes.md
for w in ffn1_step dgrad wgrad; do
  for p in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc; timeout 200 rocprofv3 --kernel-trace --pmc $p -d /tmp/pmc -- python $ROOT/tools/bench_one.py $w 10 > /tmp/pmc.log 2>&1
    echo "## $w  ($p, KiB per launch; FETCH_SIZE counts 128-B requests at 64 B on gfx950: double it)" >> $OUT/${R}_pmc_traffic_gemm_shapes.md
    python $ROOT/tools/rocpd_pmc_summary.py $(find /tmp/pmc -name "*results.db" | head -1) 2>&1 | grep -E "^\| kernel|gemm_buf" | cut -c1-300 >> $OUT/${R}_pmc_traffic_gemm_shapes.md
    grep TFLOP /tmp/pmc.log >> $OUT/${R}_pmc_traffic_gemm_shapes.md
  done
done
# SQ counters (two passes each: they do not fit one) of the fused attention kernels and of the dominant GEMM as the step launches it
P1="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"
P2="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS"
for job in "attention:bench_attn.py conformer --fused-only:attn_" "gemm:bench_one.py ffn1_step 10:gemm_buf"; do
  n=${job%%:*}; rest=${job#*:}; cmd=${rest%%:*}; pat=${rest#*:}
  : > $OUT/${R}_pmc_sq_$n.md
  for P in "$P1" "$P2"; do
    rm -rf /tmp/pmc; timeout 200 rocprofv3 --kernel-trace --pmc $P -d /tmp/pmc -- python $ROOT/tools/$cmd > /tmp/pmc.log 2>&1
    python $ROOT/tools/rocpd_pmc_summary.py $(find /tmp/pmc -name "*results.db" | head -1) 2>&1 < /dev/null | grep -E "^\| kernel|$pat" | cut -c1-400 >> $OUT/${R}_pmc_sq_$n.md
  done
done
# does a wave's VALU work overlap with its own (or a co-resident wave's) fp32 MFMAs?  (it does not: tools/ubench/mfma_valu.hip)
timeout 120 /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/mfma_valu $ROOT/tools/ubench/mfma_valu.hip > /dev/null 2>&1 && timeout 60 /tmp/mfma_valu > $OUT/${R}_ubench_mfma_valu.txt 2>&1
ls -la $OUT | grep ${R}_ | head -60

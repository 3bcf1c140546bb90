#!/bin/bash
# Round-6 artefacts in one GPU-box call:  bash tools/collect_r06.sh  -> gpurun_out/r06_*  (copy the ones to keep into profiles/)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out; R=r06
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
# headline + secondary configs (driver-style invocation)
timeout 900 python $ROOT/bench.py > $OUT/${R}_bench_fs2.json 2> $OUT/${R}_bench.err
# the replayed graph: idle time between kernels and per-kernel totals of one step
for cfg in "fs2:" "conformer:--block conformer" "c5:--prosody liu2021 --learn-alignment"; do
  n=${cfg%%:*}; a=${cfg#*:}
  rm -rf /tmp/prof_g; timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_g -- python $ROOT/bench.py $a --no-cpu-baseline --no-pcie --no-secondary --no-roofline --steps 6 --warmup 3 > /tmp/prof_g.log 2>&1
  python $ROOT/tools/graph_gaps.py $(find /tmp/prof_g -name "*results.db" | head -1) > $OUT/${R}_${n}_graph_replay_kernels.md 2>&1
done
# dominant kernel: kernel-trace summary of exactly what bench.py's roofline block launches, then PMC passes
rm -rf /tmp/prof_dom; timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_dom -- python $ROOT/tools/bench_one.py ffn1_step 60 > /tmp/prof_dom.log 2>&1
python $ROOT/tools/rocpd_summary.py $(find /tmp/prof_dom -name "*results.db" | head -1) 6 > $OUT/${R}_dominant_kernel_stats.md 2>&1
grep TFLOP /tmp/prof_dom.log >> $OUT/${R}_dominant_kernel_stats.md
: > $OUT/${R}_pmc_traffic_gemm_shapes.md
for p in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc; timeout 200 rocprofv3 --kernel-trace --pmc $p -d /tmp/pmc -- python $ROOT/tools/bench_one.py ffn1_step 30 > /tmp/pmc.log 2>&1
  echo "## ffn1_step  ($p, KiB per launch; FETCH_SIZE counts 128-B requests at 64 B on gfx950: double it)" >> $OUT/${R}_pmc_traffic_gemm_shapes.md
  python $ROOT/tools/rocpd_pmc_summary.py $(find /tmp/pmc -name "*results.db" | head -1) 2>&1 | grep -E "^\| kernel|gemm_|split_" | cut -c1-300 >> $OUT/${R}_pmc_traffic_gemm_shapes.md
  grep TFLOP /tmp/pmc.log >> $OUT/${R}_pmc_traffic_gemm_shapes.md
done
P1="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"
P2="SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS"
P3="SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM"
: > $OUT/${R}_pmc_sq_gemm.md
for P in "$P1" "$P2" "$P3"; do
  rm -rf /tmp/pmc; timeout 200 rocprofv3 --kernel-trace --pmc $P -d /tmp/pmc -- python $ROOT/tools/bench_one.py ffn1_step 30 > /tmp/pmc.log 2>&1
  python $ROOT/tools/rocpd_pmc_summary.py $(find /tmp/pmc -name "*results.db" | head -1) 2>&1 < /dev/null | grep -E "^\| kernel|gemm_" | cut -c1-400 >> $OUT/${R}_pmc_sq_gemm.md
done
# the weight-gradient plane kernel (gemm_plw.hip): same passes over the step's FFN conv weight-gradient launch
: > $OUT/${R}_pmc_plane_wgrad.md
for P in "$P1" "$P2" "$P3 SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT" "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pmc; timeout 200 rocprofv3 --kernel-trace --pmc $P -d /tmp/pmc -- python $ROOT/tools/bench_one.py wgrad_step 30 > /tmp/pmc.log 2>&1
  python $ROOT/tools/rocpd_pmc_summary.py $(find /tmp/pmc -name "*results.db" | head -1) 2>&1 < /dev/null | grep -E "^\| kernel|gemm_" | cut -c1-400 >> $OUT/${R}_pmc_plane_wgrad.md
done
echo "(FETCH_SIZE / WRITE_SIZE in KiB per launch; FETCH_SIZE counts 128-B requests at 64 B on gfx950: double it)" >> $OUT/${R}_pmc_plane_wgrad.md
# stock-torch launches that remain in one eager step, by call site
timeout 200 python $ROOT/tools/find_torch_ops.py > $OUT/${R}_torch_ops_fs2.txt 2>&1
timeout 200 python $ROOT/tools/find_torch_ops.py --block conformer > $OUT/${R}_torch_ops_conformer.txt 2>&1
timeout 300 python $ROOT/tools/find_torch_ops.py --c5 > $OUT/${R}_torch_ops_c5.txt 2>&1
# micro-benchmarks
timeout 300 python $ROOT/tools/bench_pl.py 40 > $OUT/${R}_microbench_plane_kernel.txt 2>&1
timeout 200 python $ROOT/tools/profile_gemm_shapes.py > $OUT/${R}_gemm_shapes_in_step_fs2.txt 2>&1
timeout 200 python $ROOT/tools/profile_gemm_shapes.py --block conformer > $OUT/${R}_gemm_shapes_in_step_conformer.txt 2>&1
timeout 100 python $ROOT/tools/bench_stft.py > $OUT/${R}_bench_stft.jsonl 2>/dev/null
ls -la $OUT | grep ${R}_ | head -40
# round 6: which ctts_split_planes launches remain in a step (shapes + call sites)
timeout 200 python $ROOT/tools/dbg_splits.py 2>&1 | grep -v "^0 \|amdgpu.ids" > $OUT/${R}_split_launches_fs2.txt
timeout 200 python $ROOT/tools/dbg_splits.py --conformer 2>&1 | grep -v "^0 \|amdgpu.ids" > $OUT/${R}_split_launches_conformer.txt
# the producers A/B: the same replayed step with every consumer splitting its own operand again
rm -rf /tmp/prof_g; CTTS_PRODUCER_PLANES=0 timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_g -- python $ROOT/bench.py --no-cpu-baseline --no-pcie --no-secondary --no-roofline --steps 6 --warmup 3 > /tmp/prof_g.log 2>&1
python $ROOT/tools/graph_gaps.py $(find /tmp/prof_g -name "*results.db" | head -1) > $OUT/${R}_fs2_graph_replay_kernels_producers_off.md 2>&1
ls -la $OUT | grep ${R}_ | head -60

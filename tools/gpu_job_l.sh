#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd $ROOT
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_kernels_gpu.py -m gpu -x -q -s -k "fused_accumulation_through_trainstep or conformer_b4_t1000_train_mode or c5_canonical or relpos_attention_fwd_bwd or mel_spectrogram_vs_reference" 2>&1 | grep -v "^$" | tail -30
cd /tmp
timeout 300 python $ROOT/tools/bench_staged_host_cost.py 2>&1 | grep -v amdgpu.ids
timeout 600 python $ROOT/bench.py --no-cpu-baseline > $ROOT/gpurun_out/bench_l.json 2> $ROOT/gpurun_out/bench_l.err; tail -3 $ROOT/gpurun_out/bench_l.err; python - <<'PY'
import json
d = json.loads(open('/root/repo/gpurun_out/bench_l.json').read().strip().splitlines()[-1])
print("fs2", round(d["ms_per_step"],3), "ms", round(d["value"]), "roofline", d["roofline"]["achieved"], d["roofline"]["launch_us"], "pcie", d["pcie_inclusive"]["ms_per_step"])
for s in d["secondary"]:
    print(s.get("config"), s.get("ms_per_step"), s.get("value"), s.get("step_frac_of_fp32_mfma_peak"), s.get("final_loss"), s.get("error"))
PY

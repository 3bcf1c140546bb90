"""bench.py - mel-frames/sec of one full CompTransTTS train step on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--scaling weak|strong]

N > 1 without a launcher: bench.py starts the N ranks itself (one process per GPU, like the reference's
`mp.spawn(train, nprocs=num_gpus)`, train.py:251-252).  Under `python -m torch.distributed.run --nproc-per-node N ... bench.py
--gpus N` it picks RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment instead.  Rank 0 prints ONE JSON line.

Step (train.py:102-125 of the reference): forward -> CompTransTTSLoss (step > var_start_steps, all terms active) -> backward
[-> DP: bucketed all-reduce(sum)/world of the flat gradient arena over RCCL, overlapped with the remaining backward stages] ->
clip_grad_norm_(1.0) -> Adam (Noam LR) -> zero_grad, fp32, dropout ON, synthetic LJSpeech-shaped canonical batch (SURVEY.md 8(d):
B=16, src<=128, mel<=1024, 11,992 valid frames) resident in HBM.
  --scaling weak   (default): every rank runs its own canonical batch of 16 (what the reference de facto does)
  --scaling strong : ONE global canonical batch of 16 dealt out over the ranks: --shard snake (default, length-balanced) or
                     --shard strided (r, r+N, ...: DistributedSampler, train.py:44 - 28 % more frames on rank 0 than on rank 7 at N = 8)
value = valid mel frames of all ranks per step / max-over-ranks wall time per step.

Adds `roofline` (dominant kernel: the implicit-GEMM Conv1d k=9 of the decoder FFN, fp32 MFMA peak 157.3 TFLOP/s), `cpu_baseline`
(oracle restatement of the reference's CPU PyTorch path timed on this host on a bounded sample: C2 = the same B=16 batch, and C1 =
B=4, at n = physical cores and n = 8 threads) and `pcie_inclusive` (the same step fed by the host data path: collate layout ->
one pinned buffer -> one async H2D per step -> one D2D refresh of the graph inputs) to the JSON line.

Other BASELINE configurations: --block conformer (configs[2]); --dataset VCTK (per-GPU slice of configs[3]); --learn-alignment /
--prosody liu2021 (together: configs[4] = SURVEY C5).  --no-graph launches eagerly; --no-overlap uses one blocking all-reduce.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs @ 2.4 GHz
BF16_MFMA_PEAK_TFLOPS = 2500.0    # same guide: dense v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate)
X6_TERMS = 6                      # csrc/gemm.hip gemm_x6_kernel: bf16 MFMAs per fp32-equivalent product (hi/mid/lo operand split)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: 16 utterances per GPU; strong: 16 utterances in total, sharded r::N")
    ap.add_argument("--shard", default="strided", choices=["snake", "strided"],
                    help="strong scaling: how the global batch is dealt out - strided = r::N like DistributedSampler (train.py:44), the "
                         "reference's semantics and the default (ADVICE r03); snake = length-balanced, an optimisation to be asked for; "
                         "the bench line reports max/mean valid frames per rank for both")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--no-overlap", action="store_true", help="DP: one blocking all-reduce of the whole arena after backward")
    ap.add_argument("--graph-collectives", action="store_true",
                    help="DP over RCCL: capture the whole step - backward stages, bucketed all-reduces, optimizer - as ONE hipGraph (no host "
                         "work between the stages); default: one graph per stage, eager collectives between the replays")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline", default="full", choices=["full", "primary"],
                    help="primary: only C2 at physical cores; full: C1 and C2 at physical cores and at 8 threads")
    ap.add_argument("--no-pcie", action="store_true", help="skip the second timed loop with the host data path inside")
    ap.add_argument("--no-secondary", action="store_true", help="skip the BASELINE configs[2..4] lines measured after the headline")
    ap.add_argument("--no-roofline", action="store_true", help="skip the stand-alone timing of the dominant kernel (kernel-trace runs: keeps "
                                                               "the 80 extra launches out of the per-step statistics)")
    ap.add_argument("--no-live-traffic", action="store_true", help="roofline.traffic from the committed PMC passes instead of two live "
                    "rocprofv3 --pmc passes over the dominant kernel (adds ~1 min)")
    ap.add_argument("--batch", default="canonical", choices=["canonical", "c1"])
    ap.add_argument("--block", default="transformer_fs2", choices=["transformer_fs2", "conformer"],
                    help="block_type plugin; the headline metric (BASELINE configs[1]) is transformer_fs2, conformer = configs[2]")
    ap.add_argument("--dataset", default="LJSpeech", choices=["LJSpeech", "VCTK"],
                    help="VCTK = multi-speaker yaml (external 512-d speaker embeddings, lambda_word_dur 0); with 8 utterances per GPU this "
                         "is the per-GPU slice of BASELINE configs[3] / SURVEY C4 (global batch 64 over 8 GPUs)")
    ap.add_argument("--prosody", default="none", choices=["none", "liu2021"], help="prosody_modeling.model_type (SURVEY a17)")
    ap.add_argument("--learn-alignment", action="store_true",
                    help="unsupervised durations: aligner + device MAS + ForwardSum/Bin losses (SURVEY a16); with --prosody liu2021 "
                         "this is SURVEY config C5")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------- launch
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n):
    """one process per GPU (train.py:29-35,251-252): re-run this script N times with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set;
    rank 0 inherits stdout (the JSON line), the others only stderr.  Returns the first non-zero exit code (0 if all succeeded)."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "8"))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL needs it on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is not None:
                    pending.discard(r)
                    if code != 0 and rc == 0:
                        rc = code
                        print(f"[bench] rank {r} exited with {code}; stopping the other ranks", file=sys.stderr)
                        for q in pending:
                            procs[q].terminate()          # exact PIDs we started, never a pattern
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


# ----------------------------------------------------------------------------------------------------------- measurement
def _pmc_per_launch(db_path, counter, kernel_substr="::gemm_"):
    """average over the launches of `kernel_substr` of the counter summed over its hardware instances (rocprofv3 rocpd sqlite result)"""
    import sqlite3
    db = sqlite3.connect(db_path)
    cols = [r[1] for r in db.execute("pragma table_info('counters_collection')")]
    namecol = "kernel_name" if "kernel_name" in cols else "name"
    vcol = "value" if "value" in cols else "counter_value"
    rows = db.execute(f"select dispatch_id, sum({vcol}) from counters_collection where counter_name = ? and {namecol} like ? "
                      "group by dispatch_id", (counter, f"%{kernel_substr}%")).fetchall()
    db.close()
    if not rows:
        raise RuntimeError(f"no {counter} rows for {kernel_substr} in {db_path}")
    return sum(v for _, v in rows) / len(rows), len(rows)


def measure_traffic_live(timeout=90):
    """HBM-side bytes per launch of the dominant kernel, measured IN THIS RUN: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE;
    --kernel-trace only, as the pool allows) over tools/bench_one.py ffn1_step = the same launch `measure_dominant_kernel` times, corrected
    as MI355X_MICROARCH.md prescribes for gfx950 (unit KiB; FETCH_SIZE tallies 128-B requests at 64 B -> doubled).
    Returns (bytes, source dict) or raises; the caller falls back to the committed passes."""
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        raise RuntimeError("rocprofv3 not on PATH")
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        raise RuntimeError("already running under a profiler")
    kib = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="ctts_pmc_", dir="/tmp")
        try:
            cmd = ["rocprofv3", "--kernel-trace", "--pmc", ctr, "-d", d, "--", sys.executable, os.path.join(ROOT, "tools", "bench_one.py"),
                   "ffn1_step", "10"]
            subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), timeout=timeout, check=True,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            dbs = glob.glob(os.path.join(d, "**", "*results.db"), recursive=True)
            if not dbs:
                raise RuntimeError("rocprofv3 wrote no results.db")
            kib[ctr], n = _pmc_per_launch(dbs[0], ctr)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    traffic = (2.0 * kib["FETCH_SIZE"] + kib["WRITE_SIZE"]) * 1024.0
    return traffic, {"measured": "live in this bench.py run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over "
                                 "tools/bench_one.py ffn1_step", "launches_averaged": n, "FETCH_SIZE_KiB": kib["FETCH_SIZE"],
                     "WRITE_SIZE_KiB": kib["WRITE_SIZE"], "correction": "gfx950: FETCH_SIZE doubled (128-B requests tallied at 64 B), unit KiB"}


def _time_launches(launch, iters, warm):
    """average duration of `launch` by HIP events on the launch stream (torch's current stream = the stream ctts_gemm launches on)"""
    for _ in range(warm):
        launch()
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(iters):
        launch()
    e1.record(st)
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def _time_launches_graph(launch, n=20, reps=3):
    """the same launch `n` times inside ONE hipGraph, replayed `reps` times between two HIP events: the kernel's duration without the host's
    launch path between two launches (what it costs inside the replayed train step); None when capture is not possible"""
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            launch()                          # the stream's workspace (zero-filled once) must exist BEFORE the capture
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
            for _ in range(n):
                launch()
        torch.cuda.current_stream().wait_stream(side)
        g.replay()
        st = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(reps):
            g.replay()
        e1.record(st)
        e1.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / (n * reps)
    except Exception as e:                                    # noqa: BLE001
        print(f"[bench] graph-timed launch skipped ({type(e).__name__}: {e})", file=sys.stderr)
        torch.cuda.synchronize()
        return None


def measure_dominant_kernel(dev, batch, iters=50, warm=30, live_traffic=True):
    """HIP-event timing (on the launch stream) of the three GEMMs of the decoder FFN Conv1d(256->1024, k=9) at their train-step
    arguments (M = B*Tm rows of which `valid` are not padding, N = 1024, K = 2304): forward (implicit GEMM on the activation),
    data gradient (N = 256, K = 9216) and weight gradient (TN, reduction over the rows).  `roofline` prices the FORWARD launch: it runs on
    gemm_pl_kernel, the kernel with the largest per-step total (24 launches of the fs2 step: profiles/r05_fs2_graph_replay_kernels.md);
    the other two are listed next to it (`ffn_conv`; the weight gradient runs on gemm_plw_kernel, the same arithmetic on the same planes)."""
    from ctts_amd import kernels as K
    from ctts_amd import ops as O

    B, T = batch["mels"].shape[0], batch["mels"].shape[1]
    M, cin, cout, ks = B * T, 256, 1024, 9
    x = torch.randn(B, T, cin, device=dev)
    wf = torch.randn(cout, ks * cin, device=dev) * 0.02
    bias = torch.zeros(cout, device=dev)
    out = torch.empty(B, T, cout, device=dev)
    Z = torch.empty(B, T, cout, device=dev)
    seed = torch.zeros(1, dtype=torch.int64, device=dev)
    lens = batch["mel_lens"].to(torch.int32).to(dev)          # padded-row skipping exactly as the decoder passes it (model.py FFTBlocks.run)
    valid = int(batch["mel_lens"].sum())
    bf16 = K.BF16_SPLIT >= 1
    split_us = None
    fwd_kw = dict(conv=(T, ks // 2, cin), alpha=ks ** -0.5, bias=bias, Z=Z, ldz=cout, act=K.ACT_GELU, p_drop=0.1, seed=seed, drop_offset=1,
                  row_lens=lens, row_T=T, row_halo=0)
    planes = {}
    if bf16 and K.plane_shape_ok(M, cout, ks * cin, cin):
        ap, bp = K.split_planes([x.view(M, cin), wf])
        if K.gemm_takes_planes(x, wf, out, M, cout, ks * cin, cin, ks * cin, cout, True, True, a_planes=ap, b_planes=bp, **fwd_kw):
            planes = dict(a_planes=ap, b_planes=bp)
            split_us = _time_launches(lambda: K.split_planes([x.view(M, cin)]), 20, 5) * 1e6      # the activation's split launch (the weight's is per step)
    tmap = None if planes else K.row_tile_map(lens, T, 0, M)   # device-built m-tile schedule, as ops.PadRows hands it to every layer

    def fwd():
        K.gemm(x, wf, out, M, cout, ks * cin, cin, ks * cin, cout, True, True, tile_map=tmap, **fwd_kw, **planes)
    # `warm` launches first: after the host-side pause between the timed loops and this measurement the first ~30 launches run 15 %
    # slower (594 vs 511 us, tools/dbg_dom.py) while the clocks come back up; inside the train step the GPU never idles
    dt = _time_launches(fwd, iters, warm)
    dt_graph = _time_launches_graph(fwd)      # reported next to `launch_us` (never instead of it): rounds 1 - 5 priced the eager-launch timing
    algo_flops = 2.0 * cout * ks * cin * valid          # 4,718,592 FLOP per valid frame per layer
    padded_flops = 2.0 * cout * ks * cin * M
    # ---- data gradient and weight gradient of the same layer, as ops._LinearConv.backward launches them
    dz = torch.randn(B, T, cout, device=dev) * (torch.arange(T, device=dev)[None, :, None] < lens[:, None, None])
    wd = torch.randn(cin, ks * cout, device=dev) * 0.02
    dx = torch.empty(B, T, cin, device=dev)
    dg_kw = dict(conv=(T, ks // 2, cout), alpha=ks ** -0.5, row_lens=lens, row_T=T, row_halo=ks // 2, split_overwrite=True)
    dplanes = {}
    if bf16 and K.plane_shape_ok(M, cin, ks * cout, cout):
        ap2, bp2 = K.split_planes([dz.view(M, cout), wd])
        if K.gemm_takes_planes(dz, wd, dx, M, cin, ks * cout, cout, ks * cout, cin, True, True, a_planes=ap2, b_planes=bp2, **dg_kw):
            dplanes = dict(a_planes=ap2, b_planes=bp2)
    tmap_d = None if dplanes else K.row_tile_map(lens, T, ks // 2, M)
    dt_d = _time_launches(lambda: K.gemm(dz, wd, dx, M, cin, ks * cout, cout, ks * cout, cin, True, True, tile_map=tmap_d, **dg_kw, **dplanes),
                          iters // 2, warm // 2)
    dw = torch.zeros(cout, ks * cin, device=dev)
    kmap = K.row_tile_map(lens, T, 0, M)
    sk = max(2, O._split_k_for(cout, ks * cin, M))
    wg_kw = dict(conv=(T, ks // 2, cin), conv_on_b=True, split_k=sk, alpha=ks ** -0.5, row_lens=lens, row_T=T)
    # the step's weight-gradient launch reads the plane sets made for the other two launches (x: forward, dZ: data gradient) and adds
    # into param.grad itself (csrc/gemm_plw.hip): no split and no reduce launch is part of it
    wplanes = {}
    if planes and dplanes and K.plane_wgrad_shape_ok(cout, ks * cin, M, cin):
        wplanes = dict(a_planes=dplanes["a_planes"], b_planes=planes["a_planes"])
        if not K.gemm_takes_planes(dz, x, dw, cout, ks * cin, M, cout, cin, ks * cin, False, False, **wg_kw, **wplanes):
            wplanes = {}
    dt_w = _time_launches(lambda: K.gemm(dz, x, dw, cout, ks * cin, M, cout, cin, ks * cin, False, False, tile_map=kmap, **wg_kw, **wplanes),
                          iters // 2, warm // 2)
    # traffic: measured live by two rocprofv3 --pmc passes over this same launch when rocprofv3 is available (it
    # used to be a committed constant); otherwise the committed result of the same passes (tools/collect_r05.sh) - traffic_source says which
    traffic, traffic_src = None, None
    if live_traffic:
        try:
            traffic, traffic_src = measure_traffic_live()
        except Exception as e:      # noqa: BLE001 - any profiler problem: fall back to the committed passes and say so
            traffic_src = None
            print(f"[bench] live HBM-traffic passes skipped ({type(e).__name__}: {e}); using the committed PMC passes", file=sys.stderr)
    tj = os.path.join(ROOT, "profiles", "pmc_traffic_dominant_kernel.json")
    if traffic is None and os.path.exists(tj):
        with open(tj) as f:
            tjs = json.load(f)
        traffic = tjs.get("traffic_bytes_per_launch")
        traffic_src = {"file": "profiles/pmc_traffic_dominant_kernel.json", "collected": tjs.get("collected"), "commit": tjs.get("commit"),
                       "kernel": tjs.get("kernel")}
    ach = algo_flops / dt / 1e12
    if bf16:
        # the launches run on the BF16 matrix pipe: fp32 products as six bf16 MFMA terms (exact 3-way operand split, fp32 accumulate).  The
        # pipe that bounds them is the bf16 pipe, so the algorithmic fp32 FLOPs are priced against bf16 dense peak / 6; the fraction of
        # the fp32-MFMA peak (what the launch would be bounded by on v_mfma_f32_32x32x2_f32) is reported next to it.
        peak = BF16_MFMA_PEAK_TFLOPS / X6_TERMS
        kname = "gemm_pl_kernel (bf16 planes, stream-K, LDS-DMA)" if planes else "gemm_x6_kernel (split inside the GEMM)"
        # strings <= 128 characters: the driver's record truncates longer ones (the long form lives in DESIGN.md section 5)
        extra = {"arithmetic": "fp32 in/out/accumulate; product = 6 bf16 MFMA terms of the exact hi/mid/lo split (DESIGN.md 3)",
                 "peak_definition": "2500 TFLOP/s dense bf16 MFMA / 6 terms", "frac_of_fp32_mfma_peak": ach / FP32_MFMA_PEAK_TFLOPS,
                 "executed_bf16_tflops": ach * X6_TERMS,
                 "kernel": kname + ": decoder FFN Conv1d k=9 fwd, train-step arguments, ragged rows",
                 "activation_split_launch_us": split_us}
    else:
        peak = FP32_MFMA_PEAK_TFLOPS
        extra = {"kernel": "ctts_gemm conv fwd (decoder FFN Conv1d k=9 as implicit GEMM, train-step arguments, ragged rows)"}

    def entry(t, kernel):
        a = algo_flops / t / 1e12
        return {"launch_us": t * 1e6, "achieved": a, "frac": a / peak, "frac_of_fp32_mfma_peak": a / FP32_MFMA_PEAK_TFLOPS, "kernel": kernel}
    out = {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": traffic,
           "traffic_source": traffic_src, "launch_us": dt * 1e6, "padded_tflops": padded_flops / dt / 1e12,
           "algorithmic_bytes": 4.0 * (M * cin + cout * ks * cin + 2 * valid * cout),
           "ffn_conv": {"fwd": entry(dt, "gemm_pl_kernel" if planes else ("gemm_x6_kernel" if bf16 else "gemm_sk_kernel")),
                        "dgrad": entry(dt_d, "gemm_pl_kernel" if dplanes else "gemm_sk_kernel (fp32 MFMA)"),
                        "wgrad": entry(dt_w, "gemm_plw_kernel (row-major planes of dZ and x, LDS transpose reads; += into the gradient)" if wplanes else
                                       ("gemm_x6tn_kernel" if bf16 else "fp32 tile kernels") + f" + ordered split-K sum (split_k = {sk})"),
                        "flops_each": algo_flops, "launches_per_step_each": 6}}
    out.update(extra)
    # the same three launches once more as flat scalars (the driver's record keeps scalars, not nested objects)
    out.update({"dgrad_us": dt_d * 1e6, "dgrad_frac": algo_flops / dt_d / 1e12 / peak, "wgrad_us": dt_w * 1e6,
                "wgrad_frac": algo_flops / dt_w / 1e12 / peak,
                "traffic_over_algorithmic": (traffic / out["algorithmic_bytes"]) if traffic else None,
                "traffic_live": bool(traffic_src and "measured" in traffic_src),
                "launch_us_in_graph": (dt_graph * 1e6) if dt_graph else None,
                "frac_in_graph": (algo_flops / dt_graph / 1e12 / peak) if dt_graph else None})
    return out


def _physical_cores():
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return int(n)
    except Exception:      # noqa: BLE001
        pass
    return os.cpu_count() or 1


def _cpu_train_steps(src_lens, nthreads, n_timed, warm):
    """oracle restatement of the reference's CPU PyTorch train step (dropout on) -> (median seconds per step, valid frames)"""
    import ctts_amd
    from ctts_amd.configs import get_configs
    from ctts_amd.synthetic import make_batch, as_model_args
    from oracle import restate as R           # checker / baseline only
    from oracle.loss_restate import RefLoss as CompTransTTSLoss

    pre, mc, tc = get_configs()
    torch.set_num_threads(nthreads)
    torch.manual_seed(1234)
    model = ctts_amd.CompTransTTS(pre, mc, tc)            # parameters only (reference initialisers); never run on CPU
    trainable = {k for k, p in model.named_parameters() if p.requires_grad}
    sd = {k: (v.detach().contiguous().clone().requires_grad_(True) if k in trainable else v.detach().clone())
          for k, v in model.state_dict().items()}
    params = [v for v in sd.values() if v.requires_grad]
    opt = torch.optim.Adam(params, lr=1e-4, betas=(0.9, 0.98), eps=1e-9)
    loss_fn = CompTransTTSLoss(pre, mc, tc)
    batch = make_batch(src_lens)
    args = as_model_args(batch)
    valid = int(batch["mel_lens"].sum())
    times = []
    for it in range(warm + n_timed):
        t0 = time.perf_counter()
        a = list(args); a[7] = dict(a[7])
        stats = {}
        out = R.comp_trans_tts_forward(sd, mc, pre, *a, training=True, train_dropout=True, new_stats=stats)
        inputs = [None, None] + list(a)
        inputs[9:11] = out[-2:]
        losses = loss_fn(inputs, out[:-2], 50001)
        opt.zero_grad()
        losses[0].backward()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
        with torch.no_grad():
            for k, v in stats.items():
                sd[k].copy_(v)
        if it >= warm:
            times.append(time.perf_counter() - t0)
    return sorted(times)[len(times) // 2], valid


def cpu_baseline(mode="full"):
    """Reference CPU PyTorch path, restated (oracle/restate.py; parity vs the live reference: tests/golden/reference_vs_oracle_full_size.json),
    timed on this host.  BASELINE.md section 4: C2 (the canonical B=16 batch, i.e. the SAME workload as `value`) and C1 (B=4), at
    n = physical cores and n = 8 threads.  The headline `value` of this object is C2 at the faster of the two thread counts."""
    from ctts_amd.synthetic import C1_SRC_LENS
    phys = _physical_cores()
    runs = []
    # (config, lengths, threads, timed steps, warm-up steps); every run gets a warm-up step (the first step pays allocator / thread-pool
    # start-up) and reports the MEDIAN of 5 timed steps (median of >= 5): ~26 s for C2, ~7 s for C1
    # (median of the timed steps; the n = physical-cores leg of round 4 - one cold step at 128 threads, slower than 8 threads
    #  on every host: oversubscribing the small ops hurts - is dropped)
    plan = [("C2", None, 8, 5, 1)] if mode == "primary" else \
           [("C1", C1_SRC_LENS, 8, 5, 1), ("C2", None, 8, 5, 1)]
    for name, lens, nt, n_timed, warm in plan:
        sec, valid = _cpu_train_steps(lens, nt, n_timed, warm)
        runs.append({"config": name, "threads": nt, "valid_frames": valid, "s_per_step": sec, "frames_per_s": valid / sec,
                     "timed_steps": n_timed, "warmup_steps": warm})
    c2 = [r for r in runs if r["config"] == "C2"]
    best = max(c2, key=lambda r: r["frames_per_s"])
    ratio = None
    rj = os.path.join(ROOT, "tests", "golden", "reference_vs_oracle_full_size.json")
    if os.path.exists(rj):
        with open(rj) as f:
            ratio = json.load(f).get("speed_ratio_reference_over_oracle")
    return {"value": best["frames_per_s"], "unit": "mel-frames/s", "cores": best["threads"], "kind": "port",
            "sample": f"C2 B=16 ({best['valid_frames']} frames) full train step, median of {best['timed_steps']} after {best['warmup_steps']} warm-up, "
                      f"{best['s_per_step']:.2f} s/step, {best['threads']} thr of {phys} phys cores",
            "s_per_step": best["s_per_step"], "timed_steps": best["timed_steps"], "host_physical_cores": phys,
            "runs": runs, "reference_over_port_time_ratio": ratio}


def build_step(dev, rank, world, dataset, block, prosody, learn_alignment, batch, scaling, use_graph=True, overlap=True, shard_order="strided",
               graph_collectives=False):
    """model + loss + optimizer + synthetic batch + (captured) TrainStep of one BASELINE configuration"""
    import ctts_amd
    from ctts_amd.configs import get_configs
    from ctts_amd.data import PackedBatch
    from ctts_amd.loss import CompTransTTSLoss, ScheduledOptim
    from ctts_amd.synthetic import (make_batch, make_unsup_batch, as_collated_tuple, shard, shard_valid_frames, C1_SRC_LENS,
                                    CANONICAL_SRC_LENS)
    from ctts_amd.trainer import TrainStep

    pre, mc, tc = get_configs(dataset)
    mc["block_type"] = block
    mc["prosody_modeling"]["model_type"] = prosody
    mc["duration_modeling"]["learn_alignment"] = learn_alignment
    torch.manual_seed(1234)                                   # identical init on every rank (DDP broadcast equivalent)
    model = ctts_amd.CompTransTTS(pre, mc, tc).to(dev)
    model.train()
    loss_fn = CompTransTTSLoss(pre, mc, tc).to(dev)
    optim = ScheduledOptim(model, tc, mc, 50000, capturable=True)
    src_lens = None if batch == "canonical" else C1_SRC_LENS
    extra = {}
    if dataset == "VCTK":            # C4: 8 utterances per GPU (every second canonical length), speaker embeddings ~ N(0,1)[B,512]
        src_lens = CANONICAL_SRC_LENS[rank % 2::2] if batch == "canonical" else src_lens
        extra = dict(multi_speaker=True)
    # conformer decoders crop to max_seq_len = 1000 in training (conformer.py:148-154): cap mel length
    mk = make_unsup_batch if learn_alignment else make_batch
    cap = 1000 if block == "conformer" else None
    balance = None
    if scaling == "strong":          # ONE global batch (identical on every rank before sharding), dealt out by `shard_order`
        gb = mk(src_lens, seed=1234, max_mel_cap=cap, **extra)
        batch_cpu = shard(gb, rank, world, shard_order)
        per = {o: shard_valid_frames(gb, world, o) for o in ("strided", "snake")}
        balance = {"order": shard_order, "valid_frames_per_rank": per[shard_order],
                   "max_over_mean": {o: max(v) / (sum(v) / len(v)) for o, v in per.items()}}
    else:                            # weak: a full batch per rank, different data on every rank
        batch_cpu = mk(src_lens, seed=1234 + rank, max_mel_cap=cap, **extra)
    valid_frames = int(batch_cpu["mel_lens"].sum())
    padded_frames = batch_cpu["mels"].shape[0] * batch_cpu["mels"].shape[1]
    # host data path: collate layout -> ONE pinned buffer -> ONE H2D copy; the model inputs are views of the device buffer
    collated = as_collated_tuple(batch_cpu)
    packed = PackedBatch.pack(collated)
    views, ev = packed.to_device(dev)
    torch.cuda.current_stream().wait_event(ev)
    model_args = views[2:]

    step = TrainStep(model, loss_fn, optim, model_args, world=world, use_graph=use_graph, overlap=overlap,
                     adam_step=optim.current_step,   # steady state: both the Noam schedule and Adam's bias correction at step 50,000
                     graph_collectives=graph_collectives)
    step.bind_static_buffer(packed.device_buffer)
    if prosody != "none" or learn_alignment:
        step.step_no = 100001                                 # every loss term on: bin-loss weight 1, prosody loss enabled
    mode = "eager"
    if use_graph:
        ok = 1
        try:
            step.capture()
        except Exception as e:                                # noqa: BLE001
            ok = 0
            print(f"[bench] rank {rank}: graph capture failed ({type(e).__name__}: {e})", file=sys.stderr)
        if world > 1:                                         # ranks must not diverge in launch mode: all replay graphs or all run eager
            import torch.distributed as dist
            flag = torch.tensor([ok], device=dev, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = int(flag)
        if ok:
            mode = (f"hipgraph({step.n_stages} backward stages, bucketed all-reduce between replays | clip+adam)" if step.staged
                    else "hipgraph(fwd+bwd | clip+adam)")
            if step.g_all is not None:
                mode = f"hipgraph(whole step: {step.n_stages} backward stages + bucketed all-reduces on a side stream + clip+adam in ONE graph)"
        else:
            print("[bench] running eager", file=sys.stderr)
            step.graphs = step.g_opt = step.g_all = None
    return {"step": step, "mode": mode, "batch_cpu": batch_cpu, "collated": collated, "packed": packed,
            "valid_frames": valid_frames, "padded_frames": padded_frames, "shard_balance": balance}


def _bf16_on():
    from ctts_amd import kernels as _K
    return _K.BF16_SPLIT >= 1


FWD_FLOP_PER_FRAME = {"transformer_fs2": 629.1e9 / 11992, "conformer": 454.4e9 / 11968}      # SURVEY 8(d) forward FLOP of the canonical batch


def measure_forward(dev, steps=10, warmup=3):
    """Forward-only lines (SURVEY 8(d) "also report forward-only"; the inference call is synthesize.py:95-101): for fs2 and conformer
    (a) the teacher-forced eval() forward of the canonical batch under torch.no_grad() - the fused flash-style attention pair and every
    no-grad GEMM path - replayed from a hipGraph, and (b) the FREE-RUNNING inference branch (durations from the duration predictor, length
    regulator, pitch / energy from their predictors: the branch holds a host-visible length, so it runs eagerly).  Random-init weights
    predict ~0 frames per phoneme; for (b) the duration predictor's output bias is set to log(1 + 8) so that it predicts ~8 frames per
    phoneme like the synthetic targets (stated in the line).  Returns secondary entries."""
    import math
    import ctts_amd
    from ctts_amd.configs import get_configs
    from ctts_amd.synthetic import make_batch, as_model_args
    out = []
    for block in ("transformer_fs2", "conformer"):
        try:
            pre, mc, tc = get_configs("LJSpeech")
            mc["block_type"] = block
            torch.manual_seed(1234)
            model = ctts_amd.CompTransTTS(pre, mc, tc).to(dev).eval()
            batch = make_batch(None, seed=1234, max_mel_cap=1000 if block == "conformer" else None)
            args = [a.to(dev) if torch.is_tensor(a) else a for a in as_model_args(batch)]
            args[7] = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in args[7].items()}
            valid = int(batch["mel_lens"].sum())
            fl = FWD_FLOP_PER_FRAME[block]

            def tf():
                a = list(args); a[7] = dict(a[7])
                with torch.no_grad():
                    return model(*a)
            for _ in range(warmup):
                tf()
            torch.cuda.synchronize()
            mode, run = "eager", tf
            try:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                    tf()
                torch.cuda.current_stream().wait_stream(side)
                mode, run = "hipgraph(forward)", g.replay
            except Exception as e:                                # noqa: BLE001
                print(f"[bench] forward-only {block}: capture failed ({type(e).__name__}: {e}); eager", file=sys.stderr)
                torch.cuda.synchronize()
            for _ in range(2):
                run()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                run()
            torch.cuda.synchronize()
            el = (time.perf_counter() - t0) / steps
            out.append({"config": f"forward only, {block}: teacher-forced eval() forward of the canonical batch (B=16), torch.no_grad()",
                        "key": f"fwd_eval_{'fs2' if block == 'transformer_fs2' else block}", "value": valid / el, "unit": "mel-frames/s", "ms_per_step": el * 1e3, "steps": steps, "warmup": warmup,
                        "valid_frames": valid, "launch_mode": mode, "dtype": "f32",
                        "forward_frac_of_fp32_mfma_peak": valid / el * fl / 1e12 / FP32_MFMA_PEAK_TFLOPS})
            # (b) free-running inference
            with torch.no_grad():
                model.variance_adaptor.duration_predictor.linear.bias.fill_(math.log(9.0))
            free_args = args[:4]

            def fr():
                with torch.no_grad():
                    return model(*free_args)
            for _ in range(warmup):
                o = fr()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                o = fr()
            torch.cuda.synchronize()
            el = (time.perf_counter() - t0) / steps
            frames = int(o[9].sum())                          # mel_lens of the synthesised batch
            out.append({"config": f"forward only, {block}: free-running inference (synthesize.py:95-101 call: no targets), B=16, duration-predictor "
                                  "bias = log 9 (~8 frames per phoneme, random-init weights otherwise)",
                        "key": f"fwd_infer_{'fs2' if block == 'transformer_fs2' else block}", "value": frames / el, "unit": "mel-frames/s", "ms_per_step": el * 1e3, "steps": steps, "warmup": warmup,
                        "valid_frames": frames, "launch_mode": "eager (data-dependent output length)", "dtype": "f32",
                        "forward_frac_of_fp32_mfma_peak": frames / el * fl / 1e12 / FP32_MFMA_PEAK_TFLOPS})
            del model
        except Exception as e:                                    # noqa: BLE001
            out.append({"config": f"forward only, {block}", "error": f"{type(e).__name__}: {e}"})
        torch.cuda.empty_cache()
    return out


def measure_mel_frontend(dev, iters=20, warm=30):
    """`TacotronSTFT.mel_spectrogram` (audio/stft.py:166-185) on csrc/mel.hip, driver-timed (VERDICT r05 missing #5):
    y ~ U(-0.5, 0.5) [B, 262144] -> B x 1025 frames, B = 16 (one training batch of audio) and B = 64 (a preprocessing batch); the C entry
    point alone (`ctts_mel_spectrogram`) and the API call (range assertion deferred = the bulk path), HIP events on the launch stream.
    HBM-bound: 1 KB in + 324 B out per frame, priced against 8 TB/s.  The mel BASIS is the restated librosa 0.7.2 algorithm - basis
    parity unpinned (the dependency is absent), STFT arithmetic pinned by golden G8."""
    import ctts_amd
    from ctts_amd import kernels as K
    out = []
    try:
        st = ctts_amd.TacotronSTFT(1024, 256, 1024, 80, 22050, 0, 8000).to(dev)
        g = torch.Generator().manual_seed(1)
        for B in (16, 64):
            y = (torch.rand(B, 262144, generator=g) - 0.5).to(dev)
            for path in ("kernel", "api"):
                st.use_fft, st.strict_range = True, False
                if path == "kernel":
                    ws = st._workspace()
                    run = lambda: K.mel_spectrogram_fft(y, st._window, ws, 1024, 256, 80, kmax=st._kmax)[:2]      # noqa: E731
                else:
                    run = lambda: st.mel_spectrogram(y)                                                         # noqa: E731
                mel, en = run()
                dt = _time_launches(run, iters, warm)
                frames = mel.shape[0] * mel.shape[2]
                algo = (y.numel() + mel.numel() + en.numel()) * 4
                out.append({"config": f"mel front end B={B} {path}", "value": frames / dt, "unit": "mel-frames/s", "us_per_call": dt * 1e6,
                            "frames": frames, "hbm_algorithmic_GBps": algo / dt / 1e9, "frac_of_8TBps": algo / dt / 8e12, "dtype": "f32",
                            "note": "basis parity unpinned (librosa absent); STFT pinned by G8"})
            del y
    except Exception as e:                                    # noqa: BLE001
        out.append({"config": "mel front end", "error": f"{type(e).__name__}: {e}"})
    torch.cuda.empty_cache()
    return out


SECONDARY = [   # BASELINE configs[2..4] measured in the same invocation (N = 1 only), so that every claimed configuration is driver-run
    ("configs[2] LJSpeech conformer batch=16", "conformer", dict(dataset="LJSpeech", block="conformer", prosody="none", learn_alignment=False), 113.9e6),
    ("configs[3] VCTK multi-speaker transformer_fs2, per-GPU slice (8 of 64 utterances)", "vctk_slice",
     dict(dataset="VCTK", block="transformer_fs2", prosody="none", learn_alignment=False), 157.4e6),
    ("configs[4] LJSpeech transformer_fs2 + liu2021 prosody + learn_alignment batch=16", "c5",
     dict(dataset="LJSpeech", block="transformer_fs2", prosody="liu2021", learn_alignment=True), 157.4e6),
    # the HEADLINE configuration once more with every GEMM on v_mfma_f32_32x32x2_f32 (ctts_gemm_bf16_split_enable(0)): the same-run,
    # same-box figure of the step without the bf16-split kernels
    ("configs[1] LJSpeech transformer_fs2 batch=16 with fp32 MFMAs only (bf16-split GEMM kernels switched off)", "fs2_fp32_mfma_only",
     dict(dataset="LJSpeech", block="transformer_fs2", prosody="none", learn_alignment=False, fp32_mfma_only=True), 157.4e6),
    # NOT comparable with the headline: the reference's --use_amp (train.py:59,104 amp.autocast) honoured by the plane-kernel launches -
    # operands rounded to bf16, one MFMA term (reduced precision; tolerance in tests/test_amp_gpu.py)
    ("configs[1] LJSpeech transformer_fs2 batch=16 in the AMP arithmetic (reduced precision: conv-layer GEMM operands rounded to bf16; not the headline)",
     "fs2_amp_reduced_precision", dict(dataset="LJSpeech", block="transformer_fs2", prosody="none", learn_alignment=False, amp=True), 157.4e6),
]


def measure_secondary(dev, steps=10, warmup=3):
    """same step definition and timing as the headline (inputs resident, hipGraph replay), fewer steps; FLOP per valid frame from SURVEY 8(d)"""
    out = []
    for name, key, cfg, flop_per_frame in SECONDARY:
        prev_split = None
        try:
            if cfg.get("fp32_mfma_only"):
                from ctts_amd import kernels as _K
                prev_split = _K.gemm_bf16_split_enable(False)          # the graphs captured below hold fp32-MFMA launches only
            elif cfg.get("amp"):
                from ctts_amd import kernels as _K
                prev_split = _K.gemm_bf16_split_enable("amp")          # ... the one-term launches of the plane kernels
            b = build_step(dev, 0, 1, cfg["dataset"], cfg["block"], cfg["prosody"], cfg["learn_alignment"], "canonical", "weak")
            st = b["step"]
            for _ in range(warmup):
                st()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                st()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            st.check_kernels()
            v = b["valid_frames"] * steps / el
            out.append({"config": name, "key": key, "value": v, "unit": "mel-frames/s", "ms_per_step": el / steps * 1e3, "steps": steps, "warmup": warmup,
                        "valid_frames": b["valid_frames"], "padded_frames": b["padded_frames"],
                        "step_frac_of_fp32_mfma_peak": v * flop_per_frame / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                        "final_loss": float(st.loss_val), "launch_mode": b["mode"],
                        "dtype": "bf16 operands / f32 accumulate in the conv-layer GEMMs, f32 elsewhere" if cfg.get("amp") else "f32"})
            del b, st
        except Exception as e:                                # noqa: BLE001
            out.append({"config": name, "key": key, "error": f"{type(e).__name__}: {e}"})
        finally:
            if prev_split is not None:
                from ctts_amd import kernels as _K
                _K.gemm_bf16_split_enable(prev_split)
        torch.cuda.empty_cache()
    return out


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(spawn_ranks(a.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}; launch with --gpus equal to the number of ranks")
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU path in the product)"
    if os.environ.get("CTTS_BENCH_SAME_DEVICE"):              # test hook: all ranks share cuda:0 (gloo backend)
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} wants cuda:{local_rank} but only {torch.cuda.device_count()} GPU(s) are visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("CTTS_BENCH_BACKEND", "nccl")   # "nccl" = RCCL over xGMI; "gloo" only for single-GPU smoke tests
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    from ctts_amd.data import Prefetcher
    from ctts_amd.synthetic import make_batch

    built = build_step(dev, rank, world, a.dataset, a.block, a.prosody, a.learn_alignment, a.batch, a.scaling,
                       use_graph=not a.no_graph, overlap=not a.no_overlap, shard_order=a.shard, graph_collectives=a.graph_collectives)
    step, mode, batch_cpu, collated, packed = built["step"], built["mode"], built["batch_cpu"], built["collated"], built["packed"]
    valid_frames, padded_frames = built["valid_frames"], built["padded_frames"]
    if world > 1 and rank == 0:       # evidence for a SCALE run that N ranks and RCCL were really in play
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:             # noqa: BLE001
            ver = "?"
        print(f"[bench] world_size={dist.get_world_size()} backend={dist.get_backend()} rccl={ver} "
              f"devices_visible={torch.cuda.device_count()} scaling={a.scaling}", file=sys.stderr)

    def timed(fn, n):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t)
        return el

    trace = [] if os.environ.get("CTTS_BENCH_TRACE") else None
    run_step = step
    if trace is not None:                                     # debugging aid: device-side copies of the loss of every step (no host sync)
        def run_step():
            step()
            flat = []
            for t in step.loss_terms:
                flat += [v for v in t.values()] if isinstance(t, dict) else [t]
            trace.append(torch.stack([v.detach().reshape(()).float() for v in flat]))
    for _ in range(a.warmup):
        run_step()
    elapsed = timed(run_step, a.steps)
    loss_final = float(step.loss_val)
    step.check_kernels()              # stream-K hand-off error words of every workspace (outside the timed region): raises on a failed launch
    if trace is not None and rank == 0:
        for i, t in enumerate(trace):
            print(f"[bench] step {i}: " + " ".join(f"{float(v):.4g}" for v in t), file=sys.stderr)
    if world > 1:
        vf = torch.tensor([valid_frames], device=dev, dtype=torch.float64)
        dist.all_reduce(vf)
        total_valid = float(vf)
    else:
        total_valid = float(valid_frames)
    ms_per_step = elapsed / a.steps * 1e3
    value = total_valid * a.steps / elapsed

    # second loop, PCIe inclusive: a prefetch thread packs + uploads one batch per step; the step starts with ONE D2D refresh
    pcie = None
    if not a.no_pcie:
        n_feed = a.steps + 2
        pf = Prefetcher((collated for _ in range(n_feed)), dev, depth=2)

        def fed_step():
            next(pf)
            step.feed(pf.last_buffer)
            step()
        for _ in range(2):
            fed_step()
        el2 = timed(fed_step, a.steps)
        pcie = {"value": total_valid * a.steps / el2, "unit": "mel-frames/s", "ms_per_step": el2 / a.steps * 1e3,
                "h2d_bytes_per_step": int(packed.host.numel()),
                "path": "collate-layout tuple -> PackedBatch (one pinned buffer) -> one async H2D on a copy stream (prefetch depth 2) -> "
                        "one D2D copy into the graph's static inputs"}

    # world > 1: a few extra steps with per-bucket timing events (outside the timed region) so that a SCALE run explains itself
    comm = None
    if world > 1 and step.g_all is None and step.reducer.active:
        step.reducer.start_timing()
        for _ in range(min(5, a.steps)):
            step()
        comm = step.reducer.stop_timing()
    if rank == 0:
        headline = (a.batch == "canonical" and a.block == "transformer_fs2" and a.prosody == "none" and not a.learn_alignment
                    and a.dataset == "LJSpeech")
        roof = (measure_dominant_kernel(dev, make_batch(None, seed=1234), live_traffic=(not a.no_live_traffic and world == 1))
                if (headline and not a.no_roofline) else None)
        # whole-step roofline view: 157.4 (fs2) / 113.9 (conformer) MFLOP per valid frame vs the fp32 MFMA peak
        step_tflops = (value / world) * (157.4e6 if a.block == "transformer_fs2" else 113.9e6) / 1e12
        secondary = ((measure_secondary(dev) + measure_forward(dev) + measure_mel_frontend(dev))
                     if (headline and world == 1 and not a.no_secondary) else None)
        # flat copy of the secondary results (scalars inside `config` survive the driver's record; the full entries stay under `secondary`)
        flat = {}
        for e in secondary or []:
            if "error" in e:
                flat["sec_error_" + e.get("key", "x")] = e["error"][:120]
            elif e["config"].startswith("mel front end"):
                k = e["config"].replace("mel front end ", "mel_").replace("=", "").replace(" ", "_")
                flat[f"sec_{k}_Mfps"] = round(e["value"] / 1e6, 1)
                flat[f"sec_{k}_frac_of_8TBps"] = round(e["frac_of_8TBps"], 4)
            else:
                flat[f"sec_{e['key']}_ms"] = round(e["ms_per_step"], 3)
                fr = e.get("step_frac_of_fp32_mfma_peak", e.get("forward_frac_of_fp32_mfma_peak"))
                if fr is not None:
                    flat[f"sec_{e['key']}_frac"] = round(fr, 4)
        cpu = None if (a.no_cpu_baseline or world > 1) else cpu_baseline(a.cpu_baseline)      # reported baseline: rank 0 at N = 1 only
        nb = len(batch_cpu["src_lens"])
        what = ("supervised durations, multi-speaker (per-GPU slice of BASELINE configs[3] = 64 utterances over 8 GPUs)"
                if (a.dataset == "VCTK" and not a.learn_alignment and a.prosody == "none") else
                f"supervised durations (BASELINE configs[{1 if a.block == 'transformer_fs2' else 2}])"
                if not (a.learn_alignment or a.prosody != "none") else
                f"learn_alignment={a.learn_alignment} prosody={a.prosody} (BASELINE configs[4] / SURVEY C5 family)")
        line = {
            "metric": "mel-frames/sec (train step) LJSpeech batch=16, 1/2/4/8 MI355X", "value": value, "unit": "mel-frames/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": a.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"{a.dataset} {a.block} batch={nb}/GPU seq<=128->mel<={batch_cpu['mels'].shape[1]}x80; "
                                    "full train step fwd+loss+bwd+clip+Adam, dropout on"),
                       "workload_detail": what,
                       "global_batch": nb * world if a.scaling == "weak" else 16,
                       "shard": None if a.scaling == "weak" else a.shard,
                       "valid_frames_per_gpu": valid_frames, "padded_frames_per_gpu": padded_frames, "launch_mode": mode,
                       "parallelism": f"dp{world}", "final_loss": loss_final,
                       "gemm_arithmetic": ("fp32 storage/accumulate; large GEMMs: 6 bf16 MFMA terms of the exact 3-way split (fp32-class); rest fp32 MFMA"
                                           if _bf16_on() else "v_mfma_f32_32x32x2_f32 (CTTS_X6=0)"),
                       "grad_buckets_bytes": step.reducer.bucket_bytes() if world > 1 else None,
                       "communication": comm,
                       "strong_scaling_shard": built["shard_balance"], **flat},
            "roofline": roof, "step_model_tflops_per_gpu": step_tflops,
            "step_frac_of_fp32_mfma_peak": step_tflops / FP32_MFMA_PEAK_TFLOPS,
            "pcie_inclusive": pcie,
            "secondary": secondary,
            "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

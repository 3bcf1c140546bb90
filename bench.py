"""bench.py - mel-frames/sec of one full CompTransTTS train step on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Step (train.py:102-125 of the reference): forward -> CompTransTTSLoss (step > var_start_steps, all
terms active) -> backward -> [DP: all-reduce(sum)/world of the flat gradient arena over RCCL] ->
clip_grad_norm_(1.0) -> Adam (Noam LR) -> zero_grad, fp32, dropout ON, synthetic LJSpeech-shaped
canonical batch (B=16, src<=128, mel<=1024, 11,992 valid frames) resident in HBM.
Weak scaling: every rank runs its own canonical batch (what the reference de facto does).
value = valid mel frames of all ranks / max-over-ranks wall time.

Adds `roofline` (dominant kernel: the implicit-GEMM Conv1d k=9 of the decoder FFN, fp32 MFMA peak
157.3 TFLOP/s) and `cpu_baseline` (oracle restatement of the reference's CPU PyTorch path, timed on
this host on a bounded sample) to the JSON line.

Other BASELINE configurations: --block conformer (configs[2]); --learn-alignment / --prosody liu2021
(together: configs[4] = SURVEY C5).  The step is two hipGraphs (fwd+loss+bwd | fused clip+Adam) with the
RCCL all-reduce of the flat gradient arena between them; --no-graph launches eagerly.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs @ 2.4 GHz


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
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


class TrainStep:
    """forward/loss/backward [+ gradient all-reduce] + clip + Adam, optionally as two hipGraphs
    (fwd+bwd, clip+Adam) with the RCCL all-reduce issued eagerly between them."""

    def __init__(self, model, loss_fn, optim, batch, world, use_graph):
        from ctts_amd.synthetic import as_model_args

        self.model, self.loss_fn, self.optim, self.world = model, loss_fn, optim, world
        self.args = as_model_args(batch)
        self.loss_inputs = [None, None] + list(self.args)
        self.step_no = 50001
        from ctts_amd.dp import FlatGradArena
        from ctts_amd import ops
        ops.set_grad_accumulation_fusion(True)     # kernels accumulate straight into the flat gradient arena
        if os.environ.get("CTTS_WGRAD_STREAM", "0") == "1":
            # opt-in A/B knob: wgrad GEMMs on a side stream.  Measured SLOWER on MI355X (31.1 vs 29.9 ms/step): the
            # co-scheduled GEMMs evict each other's L2 working set, which costs more than the tail rounds they fill.
            ops.set_wgrad_stream(torch.cuda.Stream())
        # flat fp32 gradient arena: p.grad are views -> one all-reduce, no bucket copies
        self.arena = FlatGradArena(model.parameters())
        self.params = self.arena.params
        self.flat_grad = self.arena.flat
        self.use_graph = use_graph
        self.fadam = None
        if os.environ.get("CTTS_TORCH_ADAM", "0") != "1":    # default: fused clip + Adam over flat arenas (csrc/optim.hip)
            from ctts_amd.dp import FlatAdam
            oc = optim._optimizer.defaults
            self.fadam = FlatAdam(self.arena, optim._optimizer.param_groups[0]["lr"], betas=tuple(oc["betas"]), eps=oc["eps"],
                                  weight_decay=oc["weight_decay"], max_norm=1.0, current_step=0)
        self.g_fb = self.g_opt = None
        self.loss_val = None

    def fwd_bwd(self):
        args = list(self.args)
        args[7] = dict(args[7])                      # the model mutates p_targets like the reference does
        out = self.model(*args, step=self.step_no)
        inputs = list(self.loss_inputs)
        inputs[9:11] = out[-2:]
        losses = self.loss_fn(inputs, out[:-2], self.step_no)
        self.flat_grad.zero_()
        losses[0].backward()
        self.loss_val = losses[0].detach()

    def reduce(self):
        self.arena.all_reduce_mean(self.world)

    def clip_and_step(self):
        if self.fadam is not None:
            self.fadam.step()
            return
        torch.nn.utils.clip_grad_norm_(self.params, 1.0, foreach=True)
        self.optim._optimizer.step()

    def capture(self):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                self.fwd_bwd(); self.reduce(); self.optim.update_learning_rate(); self.clip_and_step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.g_fb = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_fb):
            self.fwd_bwd()
        if self.fadam is not None:           # the fused clip+Adam kernels are capture-safe; torch's foreach clip + Adam (CTTS_TORCH_ADAM=1)
            self.g_opt = torch.cuda.CUDAGraph()   # mis-replays on the strided (GEMM-major) Conv1d parameters and stays eager
            with torch.cuda.graph(self.g_opt):
                self.clip_and_step()

    def __call__(self):
        self.optim.update_learning_rate()            # host scalar -> device lr tensor (outside the graphs)
        if self.g_fb is not None:
            self.g_fb.replay(); self.reduce()
            if self.g_opt is not None:
                self.g_opt.replay()
            else:
                self.clip_and_step()
        else:
            self.fwd_bwd(); self.reduce(); self.clip_and_step()
        self.step_no += 1


def measure_dominant_kernel(dev, batch, iters=20):
    """HIP-event timing (on the launch stream) of the dominant kernel at its train-step arguments:
    decoder FFN Conv1d(256->1024, k=9) as implicit GEMM, M=B*Tm rows, N=1024, K=2304."""
    from ctts_amd import kernels as K

    B, T = batch["mels"].shape[0], batch["mels"].shape[1]
    M, cin, cout, ks = B * T, 256, 1024, 9
    x = torch.randn(B, T, cin, device=dev)
    wf = torch.randn(cout, ks * cin, device=dev) * 0.02
    bias = torch.zeros(cout, device=dev)
    out = torch.empty(B, T, cout, device=dev)
    Z = torch.empty(B, T, cout, device=dev)
    seed = torch.zeros(1, dtype=torch.int64, device=dev)
    lens = batch["mel_lens"].to(torch.int32).to(dev)          # padded-row skipping exactly as the decoder passes it (model.py FFTBlocks.run)

    tmap = K.row_tile_map(lens, T, 0, M)                      # device-built m-tile schedule, as ops.PadRows hands it to every layer

    def launch():
        K.gemm(x, wf, out, M, cout, ks * cin, cin, ks * cin, cout, True, True, conv=(T, ks // 2, cin), alpha=ks ** -0.5, bias=bias,
               Z=Z, ldz=cout, act=K.ACT_GELU, p_drop=0.1, seed=seed, drop_offset=1, row_lens=lens, row_T=T, row_halo=0, tile_map=tmap)
    for _ in range(3):
        launch()
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(iters):
        launch()
    e1.record(st)
    e1.synchronize()
    dt = e0.elapsed_time(e1) * 1e-3 / iters
    valid = int(batch["mel_lens"].sum())
    algo_flops = 2.0 * cout * ks * cin * valid          # 4,718,592 FLOP per valid frame per layer
    padded_flops = 2.0 * cout * ks * cin * M
    traffic = None
    tj = os.path.join(ROOT, "profiles", "pmc_traffic_dominant_kernel.json")   # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, see profiles/
    if os.path.exists(tj):
        with open(tj) as f:
            traffic = json.load(f).get("traffic_bytes_per_launch")
    return {"bound": "mfma", "achieved": algo_flops / dt / 1e12, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": algo_flops / dt / 1e12 / FP32_MFMA_PEAK_TFLOPS, "traffic": traffic,
            "kernel": "gemm_buf_kernel<64,64,true,true,true,false> (decoder FFN Conv1d k=9 fwd as implicit GEMM, train-step arguments incl. padded-row skipping)", "launch_us": dt * 1e6,
            "padded_tflops": padded_flops / dt / 1e12}


def cpu_baseline(seconds_budget=30.0):
    """Reference CPU PyTorch path, restated (oracle/restate.py; parity vs the reference <= 2e-5 on the goldens),
    timed on this host: C1 batch (B=4, 3,032 valid frames), full train step with dropout."""
    import ctts_amd
    from ctts_amd.configs import get_configs
    from ctts_amd.synthetic import make_batch, as_model_args, C1_SRC_LENS
    from oracle import restate as R           # checker / baseline only
    from oracle.loss_restate import RefLoss as CompTransTTSLoss

    pre, mc, tc = get_configs()
    ncores = os.cpu_count() or 1
    nthreads = min(ncores, 64)
    torch.set_num_threads(nthreads)
    torch.manual_seed(1234)
    model = ctts_amd.CompTransTTS(pre, mc, tc)            # parameters only (reference initialisers); never run on CPU
    trainable = {k for k, p in model.named_parameters() if p.requires_grad}
    sd = {k: (v.detach().contiguous().clone().requires_grad_(True) if k in trainable else v.detach().clone())
          for k, v in model.state_dict().items()}
    params = [v for v in sd.values() if v.requires_grad]
    opt = torch.optim.Adam(params, lr=1e-4, betas=(0.9, 0.98), eps=1e-9)
    loss_fn = CompTransTTSLoss(pre, mc, tc)
    batch = make_batch(C1_SRC_LENS)
    args = as_model_args(batch)
    valid = int(batch["mel_lens"].sum())
    times = []
    t_start = time.perf_counter()
    for it in range(4):
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
        dt = time.perf_counter() - t0
        if it > 0:
            times.append(dt)
        if time.perf_counter() - t_start > seconds_budget and times:
            break
    med = sorted(times)[len(times) // 2]
    return {"value": valid / med, "unit": "mel-frames/s", "cores": nthreads, "kind": "port",
            "sample": f"C1 batch B=4 ({valid} valid frames), {len(times)} train steps after 1 warm-up, median {med:.2f} s/step, "
                      f"torch CPU fp32 {nthreads} threads of {ncores} cores"}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU path in the product)"
    if os.environ.get("CTTS_BENCH_SAME_DEVICE"):              # test hook: all ranks share cuda:0 (gloo backend)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("CTTS_BENCH_BACKEND", "nccl")   # "nccl" = RCCL over xGMI; "gloo" only for single-GPU smoke tests
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    import ctts_amd
    from ctts_amd.configs import get_configs
    from ctts_amd.loss import CompTransTTSLoss, ScheduledOptim
    from ctts_amd.synthetic import make_batch, make_unsup_batch, to_device, C1_SRC_LENS

    pre, mc, tc = get_configs(a.dataset)
    mc["block_type"] = a.block
    mc["prosody_modeling"]["model_type"] = a.prosody
    mc["duration_modeling"]["learn_alignment"] = a.learn_alignment
    torch.manual_seed(1234)                                   # identical init on every rank (DDP broadcast equivalent)
    model = ctts_amd.CompTransTTS(pre, mc, tc).to(dev)
    model.train()
    loss_fn = CompTransTTSLoss(pre, mc, tc).to(dev)
    optim = ScheduledOptim(model, tc, mc, 50000, capturable=True)
    src_lens = None if a.batch == "canonical" else C1_SRC_LENS
    extra = {}
    if a.dataset == "VCTK":          # C4: 8 utterances per GPU (every second canonical length), speaker embeddings ~ N(0,1)[B,512]
        from ctts_amd.synthetic import CANONICAL_SRC_LENS
        src_lens = CANONICAL_SRC_LENS[rank % 2::2] if a.batch == "canonical" else src_lens
        extra = dict(multi_speaker=True)
    # conformer decoders crop to max_seq_len = 1000 in training (conformer.py:148-154): cap mel length
    mk = make_unsup_batch if a.learn_alignment else make_batch
    batch_cpu = mk(src_lens, seed=1234 + rank, max_mel_cap=1000 if a.block == "conformer" else None, **extra)
    batch = to_device(batch_cpu, dev)
    valid_frames = int(batch_cpu["mel_lens"].sum())
    padded_frames = batch_cpu["mels"].shape[0] * batch_cpu["mels"].shape[1]

    step = TrainStep(model, loss_fn, optim, batch, world, not a.no_graph)
    if a.prosody != "none" or a.learn_alignment:
        step.step_no = 100001                                 # every loss term on: bin-loss weight 1, prosody loss enabled
    mode = "eager"
    if not a.no_graph:
        try:
            step.capture()
            mode = "hipgraph(fwd+bwd | clip+adam)"
        except Exception as e:                                # noqa: BLE001
            if rank == 0:
                print(f"[bench] graph capture failed ({type(e).__name__}: {e}); running eager", file=sys.stderr)
            step.g_fb = step.g_opt = None
    for _ in range(a.warmup):
        step()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    loss_final = float(step.loss_val)
    if world > 1:
        t = torch.tensor([elapsed], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
        vf = torch.tensor([valid_frames], device=dev, dtype=torch.float64)
        dist.all_reduce(vf)
        total_valid = float(vf)
    else:
        total_valid = float(valid_frames)
    ms_per_step = elapsed / a.steps * 1e3
    value = total_valid * a.steps / elapsed
    if rank == 0:
        headline = (a.batch == "canonical" and a.block == "transformer_fs2" and a.prosody == "none" and not a.learn_alignment
                    and a.dataset == "LJSpeech")
        roof = measure_dominant_kernel(dev, batch_cpu) if headline else None
        # whole-step roofline view: 157.4 (fs2) / 113.9 (conformer) MFLOP per valid frame vs the fp32 MFMA peak
        step_tflops = (value / world) * (157.4e6 if a.block == "transformer_fs2" else 113.9e6) / 1e12
        cpu = None if (a.no_cpu_baseline or world > 1) else cpu_baseline()      # reported baseline: rank 0 at N = 1 only
        line = {
            "metric": "mel-frames/sec (train step) LJSpeech batch=16, 1/2/4/8 MI355X", "value": value, "unit": "mel-frames/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"{a.dataset} {a.block} batch={len(batch_cpu['src_lens'])}/GPU, seq<=128 -> mel<={batch_cpu['mels'].shape[1]}x80, "
                                    + ("supervised durations, multi-speaker (per-GPU slice of BASELINE configs[3] = 64 utterances over 8 GPUs)"
                                       if (a.dataset == "VCTK" and not a.learn_alignment and a.prosody == "none") else
                                       f"supervised durations (BASELINE configs[{1 if a.block == 'transformer_fs2' else 2}])"
                                       if not (a.learn_alignment or a.prosody != "none") else
                                       f"learn_alignment={a.learn_alignment} prosody={a.prosody} (BASELINE configs[4] / SURVEY C5 family)")
                                    + "; full train step fwd+loss+bwd+clip+Adam, dropout on"),
                       "valid_frames_per_gpu": valid_frames, "padded_frames_per_gpu": padded_frames, "launch_mode": mode,
                       "parallelism": f"dp{world}", "final_loss": loss_final},
            "roofline": roof, "step_model_tflops_per_gpu": step_tflops,
            "step_frac_of_fp32_mfma_peak": step_tflops / FP32_MFMA_PEAK_TFLOPS,
            "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

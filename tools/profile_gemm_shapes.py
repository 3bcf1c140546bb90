"""Per-shape GEMM time inside the real train step (eager launches): every kernels.gemm call is bracketed by HIP events, grouped by
descriptor shape.  python tools/profile_gemm_shapes.py [--block conformer] [--steps 3]   (GPU box)
Prints one line per shape class: calls per step, average us, dense-equivalent TFLOP/s, ms per step - the place to see which launches
the step actually spends its GEMM time in (micro-benchmarks of repeated identical launches run at different clocks and cache states)."""
import argparse, collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ctts_amd
from ctts_amd import kernels as K
from ctts_amd.configs import get_configs
from ctts_amd.data import PackedBatch
from ctts_amd.loss import CompTransTTSLoss, ScheduledOptim
from ctts_amd.synthetic import make_batch, as_collated_tuple
from ctts_amd.trainer import TrainStep

ap = argparse.ArgumentParser()
ap.add_argument("--block", default="transformer_fs2")
ap.add_argument("--steps", type=int, default=3)
a = ap.parse_args()
dev = torch.device("cuda", 0)
pre, mc, tc = get_configs("LJSpeech")
mc["block_type"] = a.block
torch.manual_seed(1234)
model = ctts_amd.CompTransTTS(pre, mc, tc).to(dev).train()
loss_fn = CompTransTTSLoss(pre, mc, tc).to(dev)
optim = ScheduledOptim(model, tc, mc, 50000, capturable=True)
batch = make_batch(None, seed=1234, max_mel_cap=1000 if a.block == "conformer" else None)
packed = PackedBatch.pack(as_collated_tuple(batch))
views, ev = packed.to_device(dev)
torch.cuda.current_stream().wait_event(ev)
step = TrainStep(model, loss_fn, optim, views[2:], use_graph=False, adam_step=optim.current_step)
for _ in range(2):
    step()
torch.cuda.synchronize()

records = []
orig = K.gemm


def epi_sig(kw):
    """epilogue signature: b(ias) a<act> z d(rop) r(esidual) s(rowscale) B(epi_bwd) E(softmax bwd)"""
    sig = ""
    sig += "b" if kw.get("bias") is not None else ""
    sig += f"a{kw['act']}" if kw.get("act") else ""
    sig += "z" if kw.get("Z") is not None else ""
    sig += "d" if kw.get("p_drop", 0.0) > 0 else ""
    sig += "r" if kw.get("R") is not None else ""
    sig += "s" if kw.get("rowscale") is not None else ""
    sig += "B" if kw.get("epi_bwd") else ""
    sig += "E" if kw.get("E") is not None else ""
    return sig or "-"


def timed_gemm(A, B, Cout, M, N, Kd, lda, ldb, ldc, a_kc=True, b_kc=True, **kw):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    r = orig(A, B, Cout, M, N, Kd, lda, ldb, ldc, a_kc, b_kc, **kw)
    e.record()
    nb = kw.get("nb0", 1) * kw.get("nb1", 1)
    key = (M, N, Kd, "NT" if (a_kc and b_kc) else ("NN" if a_kc else "TN"), "conv" if kw.get("conv") else "-", kw.get("split_k", 1), nb,
           "ragged" if kw.get("row_lens") is not None else ("lens" if kw.get("lens") is not None else "-"), epi_sig(kw))
    records.append((key, s, e))
    return r


K.gemm = timed_gemm
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(a.steps):
    step()
t1.record()
torch.cuda.synchronize()
agg = collections.defaultdict(list)
for key, s, e in records:
    agg[key].append(s.elapsed_time(e) * 1e3)
tot = sum(sum(v) for v in agg.values()) / a.steps / 1e3
print(f"# {a.block}: eager step {t0.elapsed_time(t1)/a.steps:.2f} ms (with per-GEMM events), GEMM launches {len(records)//a.steps} per step, {tot:.2f} ms per step"
      f"  [CTTS_SK={os.environ.get('CTTS_SK', '1')}]")
print(f"{'M':>6} {'N':>5} {'K':>6} lay conv sk nb  pad    | calls/step  avg us   dense TF  ms/step")
for key, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    M, N, Kd, lay, cv, sk, nb, rg, sig = key
    avg = sum(v) / len(v)
    tf = 2.0 * M * N * Kd * nb / (avg * 1e-6) / 1e12
    ms = sum(v) / a.steps / 1e3
    if ms < 0.02:
        continue
    print(f"{M:6d} {N:5d} {Kd:6d} {lay:>3} {cv:>4} {sk:2d} {nb:3d} {rg:>6} | {len(v)/a.steps:8.1f} {avg:9.1f} {tf:9.1f} {ms:8.3f}  {sig}")

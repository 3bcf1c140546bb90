"""debug: fused attention under hipGraph replay vs eager (loss trajectories, dropout off / on)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctts_amd
from ctts_amd import ops
from ctts_amd.configs import get_configs
from ctts_amd.loss import CompTransTTSLoss, ScheduledOptim
from ctts_amd.synthetic import make_batch, to_device, as_model_args
from ctts_amd.trainer import TrainStep
DEV = torch.device("cuda:0")


def run_bench_like(fused, packed, adam_step, n=25):
    from ctts_amd.data import PackedBatch
    from ctts_amd.synthetic import as_collated_tuple
    ops.set_fused_attention(fused)
    torch.manual_seed(1234)
    pre, mc, tc = get_configs()
    model = ctts_amd.CompTransTTS(pre, mc, tc).to(DEV)
    model.train()
    loss_fn = CompTransTTSLoss(pre, mc, tc).to(DEV)
    optim = ScheduledOptim(model, tc, mc, 50000, capturable=True)
    b = make_batch(None, seed=1234)
    if packed:
        pk = PackedBatch.pack(as_collated_tuple(b))
        views, ev = pk.to_device(DEV)
        torch.cuda.current_stream().wait_event(ev)
        args = views[2:]
    else:
        args = as_model_args(to_device(b, DEV))
    step = TrainStep(model, loss_fn, optim, args, use_graph=True, adam_step=adam_step)
    step.capture()
    ls = []
    nosync = "--nosync" in sys.argv
    for _ in range(n):
        step()
        ls.append(step.loss_val.clone() if nosync else round(float(step.loss_val), 3))
    torch.cuda.synchronize()
    ls = [round(float(x), 3) for x in ls]
    print(f"benchlike fused={int(fused)} packed={int(packed)} adam_step={adam_step} nosync={int(nosync)}: {ls}", flush=True)


def run(fused, graph, dropout, n=8, lens=None):
    ops.set_fused_attention(fused)
    torch.manual_seed(1234)
    pre, mc, tc = get_configs()
    model = ctts_amd.CompTransTTS(pre, mc, tc).to(DEV)
    if not dropout:
        for m in model.modules():
            if hasattr(m, "dropout"):
                m.dropout = 0.0
    model.train()
    loss_fn = CompTransTTSLoss(pre, mc, tc).to(DEV)
    optim = ScheduledOptim(model, tc, mc, 50000, capturable=True)
    step = TrainStep(model, loss_fn, optim, as_model_args(to_device(make_batch(lens), DEV)), use_graph=graph)
    if graph:
        step.capture(warmup=2)
    else:
        for _ in range(2):
            step.optim.update_learning_rate(); step._eager()
    ls = []
    for _ in range(n):
        step()
        ls.append(round(float(step.loss_val), 4))
    print(f"fused={int(fused)} graph={int(graph)} dropout={int(dropout)} qsplit={os.environ.get('CTTS_ATTN_Q_SPLIT','auto')} lens={'canon' if lens is None else lens}: {ls}", flush=True)


if __name__ == "__main__":
    if "--benchlike" in sys.argv:
        run_bench_like(True, True, 50000)
        run_bench_like(True, False, 50000)
        run_bench_like(False, True, 50000)
        sys.exit(0)
    drop = "--dropout" in sys.argv
    n = 26 if drop else 8
    for lens in (None, [60, 41, 33, 17]):
        run(True, False, drop, n=n, lens=lens)
        run(True, True, drop, n=n, lens=lens)
        run(False, True, drop, n=n, lens=lens)

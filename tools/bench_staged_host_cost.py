"""Host-side cost of the STAGED train step (4 backward-stage hipGraphs + the optimizer graph, launched one after the other from Python,
with the bucket reducer's calls in between) against the monolithic step (1 + 1 graphs), at world = 1 where no collective runs: what is
left is pure launch overhead.  Measured at the strong-scaling shard of N = 8 (2 utterances per rank: the shortest step a rank ever
runs) and at the full canonical batch.   python tools/bench_staged_host_cost.py   (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ctts_amd
from ctts_amd.configs import get_configs
from ctts_amd.loss import CompTransTTSLoss, ScheduledOptim
from ctts_amd.synthetic import make_batch, shard, to_device, as_model_args
from ctts_amd.trainer import TrainStep

dev = torch.device("cuda", 0)


def run(batch_cpu, staged, steps=60):
    pre, mc, tc = get_configs()
    torch.manual_seed(1234)
    model = ctts_amd.CompTransTTS(pre, mc, tc).to(dev).train()
    loss_fn = CompTransTTSLoss(pre, mc, tc).to(dev)
    optim = ScheduledOptim(model, tc, mc, 50000, capturable=True)
    st = TrainStep(model, loss_fn, optim, as_model_args(to_device(batch_cpu, dev)), world=1, force_staged=staged, adam_step=optim.current_step)
    st.capture()
    for _ in range(5):
        st()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        st()
    t_issue = time.perf_counter() - t0            # host time to ISSUE the steps (GPU still running)
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    return t_all / steps * 1e3, t_issue / steps * 1e3, st.n_stages


gb = make_batch()
for name, b in (("strong-scaling shard of N=8 (2 utterances, snake)", shard(gb, 0, 8, "snake")), ("canonical batch (16 utterances)", gb)):
    mono = run(b, False)
    stag = run(b, True)
    print(f"{name}: monolithic {mono[0]:.3f} ms/step (host issue {mono[1]:.3f} ms) | staged x{stag[2]} {stag[0]:.3f} ms/step (host issue {stag[1]:.3f} ms) | "
          f"staging costs {100 * (stag[0] / mono[0] - 1):+.1f} %", flush=True)

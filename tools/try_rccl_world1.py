"""One-rank RCCL on a single GPU: the only way to execute the RCCL code path of the DP step on a 1-GPU box.
backend "nccl" = RCCL, world_size 1; the bucketed all-reduces (ReduceOp.AVG, side stream) run (a) eagerly between eager stages,
(b) eagerly between the replays of the 4 stage graphs, (c) captured INSIDE one whole-step hipGraph (TrainStep graph_collectives), and
(d, e) the same through the library's own C-ABI collective (ctts_comm_create / ctts_allreduce_mean) eagerly and inside the whole-step
graph - and must reproduce the loss trajectory of the step without any collective bit for bit (the average over one rank is the identity).

    python tools/try_rccl_world1.py            (prints one line per mode; exit code 1 on a mismatch)"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctts_amd                                                           # noqa: E402
from ctts_amd.configs import get_configs                                  # noqa: E402
from ctts_amd.synthetic import make_batch, to_device, as_model_args       # noqa: E402

DEV = torch.device("cuda:0")


def run(mode, n=4):
    from ctts_amd.loss import CompTransTTSLoss, ScheduledOptim
    from ctts_amd.trainer import TrainStep
    torch.manual_seed(1234)
    pre, mc, tc = get_configs()
    model = ctts_amd.CompTransTTS(pre, mc, tc).to(DEV)
    model.train()
    loss_fn, optim = CompTransTTSLoss(pre, mc, tc).to(DEV), ScheduledOptim(model, tc, mc, 50000, capturable=True)
    batch = to_device(make_batch([60, 41, 33, 17], 8, seed=3), DEV)
    kw = dict(world=1)
    if mode == "none":
        kw.update(use_graph=False, force_staged=True)
    elif mode == "eager":
        kw.update(use_graph=False, always_reduce=True)
    elif mode == "stage-graphs":
        kw.update(use_graph=True, always_reduce=True, graph_collectives=False)
    elif mode == "whole-step-graph":
        kw.update(use_graph=True, always_reduce=True, graph_collectives=True)
    else:
        # the library's own C-ABI collective (include/ctts.h ctts_comm_create / ctts_allreduce_mean, csrc/comm.hip) instead of
        # torch.distributed: eager between eager stages ("abi-eager") and captured inside the whole-step graph ("abi-whole-step-graph")
        os.environ["CTTS_ABI_COLLECTIVE"] = "1"
        kw.update(use_graph=mode != "abi-eager", always_reduce=True, graph_collectives=mode != "abi-eager")
    try:
        step = TrainStep(model, loss_fn, optim, as_model_args(batch), **kw)
    finally:
        os.environ.pop("CTTS_ABI_COLLECTIVE", None)
    if mode.startswith("abi"):
        assert step.reducer._abi_comm is not None and step.reducer._abi_comm.world == 1

    if kw["use_graph"]:
        step.capture(warmup=2)
        assert (step.g_all is not None) == (mode.endswith("whole-step-graph"))
    else:
        for _ in range(2):
            step.optim.update_learning_rate()
            step._eager()
    losses = []
    for _ in range(n):
        step()
        losses.append(float(step.loss_val))
    torch.cuda.synchronize()
    return losses, step.fadam.flat_param.clone(), step.reducer.active


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        ver = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:                                      # noqa: BLE001
        ver = "?"
    print(f"backend={dist.get_backend()} rccl={ver} world={dist.get_world_size()}")
    ref = run("none")
    print("no collectives      ", ref[0])
    ok = True
    for mode in ("eager", "stage-graphs", "whole-step-graph", "abi-eager", "abi-whole-step-graph"):
        try:
            got = run(mode)
            same = got[0] == ref[0] and torch.equal(got[1], ref[1])
            print(f"{mode:20s}", got[0], "reducer active:", got[2], "identical to no-collective run:", same)
            ok &= same and got[2]
        except Exception as e:                             # noqa: BLE001
            print(f"{mode:20s} FAILED: {type(e).__name__}: {str(e)[:300]}")
            ok = False
    sys.stdout.flush()
    try:
        dist.destroy_process_group()
    except Exception as e:                                 # noqa: BLE001
        print(f"destroy_process_group: {type(e).__name__}: {e}")
    sys.stdout.flush()
    # leave without the interpreter's teardown: RCCL's finalisers abort once in a while on this stack AFTER everything was compared and
    # printed (one of six runs of the round-5 suite returned non-zero with all three modes identical) - the verdict is `ok`, not the exit path
    os._exit(0 if ok else 1)


if __name__ == "__main__":
    main()

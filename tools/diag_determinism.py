"""Round-4 root-cause tool for tests/test_dp_gpu.py::test_hipgraph_replay_matches_eager_training.

Runs the SAME train-step trajectory (same seed, same batch) several times in one process - eager twice, hipGraph replay twice - and
prints, for every pair: the loss trajectories, whether they are bit-identical, and after the FIRST measured step the parameters whose
gradient differs between the two runs (max abs difference, relative to the tensor's max).  eager != eager  =>  run-to-run
non-determinism of the kernels (fp32 atomics);  eager == eager but graph != eager  =>  a capture / replay defect.

    python tools/diag_determinism.py [--c5 0|1] [--block transformer_fs2|conformer] [--steps 5]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctts_amd                                                           # noqa: E402
from ctts_amd.configs import get_configs                                  # noqa: E402
from ctts_amd.synthetic import make_batch, make_unsup_batch, to_device, as_model_args   # noqa: E402

DEV = torch.device("cuda:0")


def make(c5, block):
    from ctts_amd.loss import CompTransTTSLoss, ScheduledOptim
    torch.manual_seed(1234)
    pre, mc, tc = get_configs()
    mc["block_type"] = block
    if c5:
        mc["prosody_modeling"]["model_type"] = "liu2021"
        mc["duration_modeling"]["learn_alignment"] = True
    model = ctts_amd.CompTransTTS(pre, mc, tc).to(DEV)
    model.train()
    return model, CompTransTTSLoss(pre, mc, tc).to(DEV), ScheduledOptim(model, tc, mc, 50000, capturable=True)


def trajectory(c5, block, use_graph, n, lens, tmel):
    from ctts_amd.trainer import TrainStep
    model, loss_fn, optim = make(c5, block)
    cap = 1000 if block == "conformer" else None
    batch = to_device((make_unsup_batch if c5 else make_batch)(lens, tmel, seed=3, max_mel_cap=cap), DEV)
    step = TrainStep(model, loss_fn, optim, as_model_args(batch), world=1, use_graph=use_graph)
    if c5:
        step.step_no = 100001
    snaps = []
    if use_graph:
        step.capture(warmup=2)
    else:
        for _ in range(2):
            step.optim.update_learning_rate()
            step._eager()
    torch.cuda.synchronize()
    snaps.append(("after warm-up: grads", step.flat_grad.clone()))
    snaps.append(("after warm-up: params", step.fadam.flat_param.clone()))
    losses = []
    for i in range(n):
        step()
        losses.append(float(step.loss_val))
        if i == 0:
            torch.cuda.synchronize()
            snaps.append(("step 1: grads", step.flat_grad.clone()))
    torch.cuda.synchronize()
    snaps.append(("final params", step.fadam.flat_param.clone()))
    return losses, snaps, step


def compare(tag, a, b, step):
    la, sa = a
    lb, sb = b
    same = all(x == y for x, y in zip(la, lb))
    print(f"--- {tag}: losses bit-identical = {same}")
    print("   A", la)
    print("   B", lb)
    for (name, ta), (_, tb) in zip(sa, sb):
        d = (ta - tb).abs()
        nz = int((d > 0).sum())
        print(f"   {name}: bit-identical = {nz == 0}; differing elements {nz} / {d.numel()}, max abs diff {float(d.max()):.3e} (|max| {float(ta.abs().max()):.3e})")
        if nz and "grads" in name:
            rows = []
            ar = step.arena
            for pname, off, prm in zip(ar.names, ar.offsets, ar.params):
                num = prm.numel()
                dd = d[off:off + num]
                if float(dd.max()) > 0:
                    rows.append((float(dd.max()) / max(float(ta[off:off + num].abs().max()), 1e-30), float(dd.max()), pname, num))
            rows.sort(reverse=True)
            for r in rows[:12]:
                print(f"        rel {r[0]:.2e} abs {r[1]:.2e}  {r[2]} [{r[3]}]")
            print(f"        ({len(rows)} tensors differ)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--c5", type=int, default=1)
    ap.add_argument("--block", default="transformer_fs2")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--canonical", type=int, default=0, help="1: the B=16 canonical batch instead of the 4-utterance test batch")
    ap.add_argument("--graph", type=int, default=1)
    a = ap.parse_args()
    if a.canonical:
        from ctts_amd.synthetic import CANONICAL_SRC_LENS
        lens, tmel = CANONICAL_SRC_LENS, 8
    else:
        lens, tmel = [60, 41, 33, 17], 8
    c5 = bool(a.c5)
    e1 = trajectory(c5, a.block, False, a.steps, lens, tmel)
    e2 = trajectory(c5, a.block, False, a.steps, lens, tmel)
    compare("eager vs eager", e1[:2], e2[:2], e1[2])
    if a.graph:
        g1 = trajectory(c5, a.block, True, a.steps, lens, tmel)
        g2 = trajectory(c5, a.block, True, a.steps, lens, tmel)
        compare("graph vs graph", g1[:2], g2[:2], e1[2])
        compare("eager vs graph", e1[:2], g1[:2], e1[2])


if __name__ == "__main__":
    main()

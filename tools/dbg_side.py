"""debug: loss terms of the C5 step, eager vs hipGraph replay (side-stream CTC loss)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ctts_amd import kernels as K
from tests.test_fullsize_gpu import _step
stash = {}
orig = K.forward_sum_fwd
def spy(a, in32, out32, blank):
    stash["a"], stash["in32"], stash["out32"] = a, in32, out32
    stash["chk"] = a.double().sum().reshape(1)
    stash["stream"] = torch.cuda.current_stream().cuda_stream
    nll, lse, alpha = orig(a, in32, out32, blank)
    stash["nll"] = nll
    return nll, lse, alpha
K.forward_sum_fwd = spy
def terms(step):
    out = []
    for t in step.loss_terms:
        out += [float(v.detach()) for v in t.values()] if isinstance(t, dict) else [float(t.detach())]
    return [round(v, 4) for v in out]
for use_graph in (False, True):
    step, model, batch = _step(c5=True, use_graph=use_graph)
    if use_graph:
        step.capture(warmup=2)
    else:
        for _ in range(2):
            step.optim.update_learning_rate(); step._eager()
    for i in range(3):
        step()
        torch.cuda.synchronize()
        print("graph" if use_graph else "eager", i, "ctc", terms(step)[11], "nll", [round(v, 3) for v in stash["nll"].tolist()], "in/out", stash["in32"].tolist(), stash["out32"].tolist(),
              "chk(side)", float(stash["chk"]), "chk(now)", float(stash["a"].double().sum()), "stream", hex(stash["stream"]))

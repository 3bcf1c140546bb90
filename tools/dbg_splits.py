"""which ctts_split_planes launches remain in one eager canonical fs2 step (shapes + call site)"""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ctts_amd
from ctts_amd import kernels as K
from ctts_amd.configs import get_configs
from ctts_amd.loss import CompTransTTSLoss, ScheduledOptim
from ctts_amd.trainer import TrainStep
from ctts_amd.synthetic import make_batch, to_device, as_model_args
dev = torch.device("cuda:0")
pre, mc, tc = get_configs()
if "--conformer" in sys.argv:
    mc["block_type"] = "conformer"
torch.manual_seed(1234)
model = ctts_amd.CompTransTTS(pre, mc, tc).to(dev).train()
loss_fn, optim = CompTransTTSLoss(pre, mc, tc).to(dev), ScheduledOptim(model, tc, mc, 50000, capturable=True)
batch = to_device(make_batch(None, seed=1234, max_mel_cap=1000 if "--conformer" in sys.argv else None), dev)
step = TrainStep(model, loss_fn, optim, as_model_args(batch), world=1, use_graph=False)
for _ in range(2):
    step.optim.update_learning_rate(); step._eager()
orig = K.split_planes
log = []
def spy(mats):
    st = traceback.extract_stack(limit=5)
    log.append(([tuple(m.shape) for m in mats], " <- ".join(f"{f.name}:{f.lineno}" for f in st[:-1][::-1])))
    return orig(mats)
K.split_planes = spy
step.optim.update_learning_rate(); step._eager()
torch.cuda.synchronize()
for shapes, site in log:
    print(len(shapes), shapes[:3], site)

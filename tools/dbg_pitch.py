"""debug: ctts_cwt_pitch vs the stock-torch chain on the G3 inference case"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ctts_amd
from ctts_amd import kernels as K, model as M
from tests.util import load_golden, closed_form_sd, batch_from_golden
from tests.test_model_gpu import build, args_from
g = load_golden("g3_fs2_infer")
m, _ = build(sd=closed_form_sd())
m.eval()
cap = {}
orig = K.cwt_pitch
def spy(spec, mean, std, scale, **kw):
    out = orig(spec, mean, std, scale, **kw)
    cap.update(spec=spec.clone(), mean=mean.clone(), std=std.clone(), scale=scale, kw=kw, out=out)
    return out
K.cwt_pitch = spy
with torch.no_grad():
    out = m(*args_from(batch_from_golden(g)), p_control=1.1, e_control=0.9, d_control=2.0)
spec, mean, std = cap["spec"], cap["mean"], cap["std"]
print("spec", tuple(spec.shape), "scale", cap["scale"], {k: v for k, v in cap["kw"].items() if not torch.is_tensor(v)})
f0, den, ids = cap["out"]
ref_f0 = M.cwt2f0_norm(spec[:, :, :10], mean, std * cap["scale"], spec.shape[1], cap["kw"]["eps"])
uv = spec[:, :, -1] > 0
ref_den = torch.where(uv, torch.zeros_like(ref_f0), 2 ** ref_f0)
print("kernel vs torch chain: f0", float((f0 - ref_f0).abs().max()), "den", float((den - ref_den).abs().max()), "ids", int((ids - M.f0_to_coarse(ref_den)).abs().max()))
gd = torch.from_numpy(g["out.f0_denorm"]).to(den.device)
print("golden shape", tuple(gd.shape), "kernel vs golden", float((den - gd).abs().max()), "torch chain vs golden", float((ref_den - gd).abs().max()))
rec = (spec[:, :, :10] * (torch.arange(10, device=spec.device) + 3.5) ** -2.5).sum(-1)
print("rec std per utterance", rec.std(-1).tolist(), "mean", rec.mean(-1).tolist())
i = (den - gd).abs().argmax(); b, t = divmod(int(i), den.shape[1]); print("worst at", b, t, float(den[b, t]), float(gd[b, t]), float(ref_den[b, t]))

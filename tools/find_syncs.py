"""Debug helper (GPU box): list host<->device synchronisation points of one C5 train step (they break hipGraph capture)."""
import os, sys, warnings, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ctts_amd
from ctts_amd.configs import get_configs
from ctts_amd.loss import CompTransTTSLoss
from ctts_amd.synthetic import make_unsup_batch, to_device, as_model_args

pre, mc, tc = get_configs()
mc["prosody_modeling"]["model_type"] = "liu2021"
mc["duration_modeling"]["learn_alignment"] = True
dev = "cuda"
m = ctts_amd.CompTransTTS(pre, mc, tc).to(dev).train()
L = CompTransTTSLoss(pre, mc, tc).to(dev)
bc = make_unsup_batch([40, 33], 8)
L.host_lens = (bc["src_lens"].tolist(), bc["mel_lens"].tolist())
b = to_device(bc, dev)
args = list(as_model_args(b))


def step():
    a = list(args); a[7] = dict(a[7])
    out = m(*a, step=100001)
    inp = [None, None] + a
    inp[9:11] = out[-2:]
    loss = L(inp, out[:-2], 100001)[0]
    loss.backward()


step()
torch.cuda.synchronize()
seen = set()


def show(message, category, filename, lineno, file=None, line=None):
    st = [f for f in traceback.extract_stack() if "/repo/" in f.filename and "find_syncs" not in f.filename]
    key = tuple((f.filename.split("/repo/")[-1], f.lineno) for f in st[-3:])
    if key not in seen:
        seen.add(key)
        print("SYNC:", str(message)[:80], "<-", " <- ".join(f"{a}:{b}" for a, b in reversed(key)))


warnings.showwarning = show
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode(1)
step()
torch.cuda.set_sync_debug_mode(0)
print("done;", len(seen), "distinct sync sites")

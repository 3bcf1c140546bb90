"""Which stock-torch (aten) kernels does one train step still launch, and from which line of the package?  One eager step under
torch.profiler (with_stack), aggregated by aten op + innermost package frame.   python tools/find_torch_ops.py [--block conformer]"""
import argparse, collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import ctts_amd
from ctts_amd.configs import get_configs
from ctts_amd.data import PackedBatch
from ctts_amd.loss import CompTransTTSLoss, ScheduledOptim
from ctts_amd.synthetic import make_batch, as_collated_tuple
from ctts_amd.trainer import TrainStep

ap = argparse.ArgumentParser()
ap.add_argument("--block", default="transformer_fs2")
ap.add_argument("--shapes", action="store_true")
ap.add_argument("--c5", action="store_true", help="BASELINE configs[4]: liu2021 prosody + learn_alignment (built through bench.build_step)")
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
if a.c5:
    import bench
    step = bench.build_step(dev, 0, 1, "LJSpeech", "transformer_fs2", "liu2021", True, "canonical", "weak", use_graph=False)["step"]
for _ in range(2):
    step()
torch.cuda.synchronize()
import traceback
from torch.utils._python_dispatch import TorchDispatchMode

agg = collections.defaultdict(int)
VIEW = ("view", "reshape", "transpose", "permute", "expand", "slice", "select", "unsqueeze", "squeeze", "as_strided", "detach", "alias",
        "t.default", "_unsafe_view", "split", "unbind", "narrow", "empty", "size", "stride", "is_", "_local_scalar", "item", "numel", "dim",
        "storage_offset", "sym_", "lift_fresh", "new_empty")


class Rec(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(v in name for v in VIEW):
            fr = "?"
            for f in reversed(traceback.extract_stack()):
                if "comprehensive-transformer-tts_amd/" in f.filename:
                    fr = f"{f.filename.split('comprehensive-transformer-tts_amd/')[-1]}:{f.lineno} {f.name}"
                    break
            shp = ""
            if "backward_stages" in fr or "--shapes" in sys.argv:
                shp = " " + ",".join(str(tuple(x.shape)) for x in args if torch.is_tensor(x))
            agg[(name, fr + shp)] += 1
        return func(*args, **(kwargs or {}))


with torch.autograd.set_multithreading_enabled(False), Rec():
    step()
torch.cuda.synchronize()
print(f"# {a.block}: {sum(agg.values())} non-view aten calls in one eager step (forward + backward + optimizer), by call site")
for (name, frame), n in sorted(agg.items(), key=lambda kv: (-kv[1], kv[0])):
    print(f"{n:4d}x  {name:34s} {frame}")

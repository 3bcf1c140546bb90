"""One data-parallel rank of tests/test_dp_gpu.py (not a test module): builds the real CompTransTTS, takes its shard of a small
global batch, runs `n_steps` steps of trainer.TrainStep (staged backward + bucketed all-reduce) and saves the flat parameter arena.

Backend (`pick_backend`): when the box has at least WORLD_SIZE GPUs every rank takes its own (`cuda:{LOCAL_RANK or RANK}`) and the group
is "nccl" = RCCL over xGMI - the configuration the reference's train.py:29-35,58 runs; on a 1-GPU box all ranks share cuda:0 and gloo
stands in (RCCL cannot put two ranks on one device).  CTTS_TEST_BACKEND=gloo|nccl forces one.

    RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT from the environment;  argv: out_path block n_steps use_graph
"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build(block, dev):
    import ctts_amd
    from ctts_amd.configs import get_configs
    from ctts_amd.loss import CompTransTTSLoss, ScheduledOptim
    pre, mc, tc = get_configs()
    mc["block_type"] = block
    torch.manual_seed(1234)
    model = ctts_amd.CompTransTTS(pre, mc, tc).to(dev)
    for m in model.modules():                        # dropout off: the comparison with the single-process step must be exact
        if hasattr(m, "dropout"):
            m.dropout = 0.0
    model.train()
    return model, CompTransTTSLoss(pre, mc, tc).to(dev), ScheduledOptim(model, tc, mc, 50000, capturable=True)


def global_batch():
    from ctts_amd.synthetic import make_batch
    return make_batch([40, 33, 21, 12], 6, seed=5)


def pick_backend(rank, world):
    """-> (backend, device): RCCL with one GPU per rank when the box has them, else gloo with every rank on cuda:0"""
    forced = os.environ.get("CTTS_TEST_BACKEND", "")
    n_gpu = torch.cuda.device_count()
    if forced == "nccl" or (forced != "gloo" and n_gpu >= world and world > 1):
        if n_gpu < world:
            raise RuntimeError(f"CTTS_TEST_BACKEND=nccl needs {world} GPUs, this box has {n_gpu}")
        local = int(os.environ.get("LOCAL_RANK", rank))
        return "nccl", torch.device(f"cuda:{local}")
    return "gloo", torch.device("cuda:0")


def main():
    out_path, block, n_steps, use_graph = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4] == "1"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend, dev = pick_backend(rank, world)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    print(f"rank {rank}/{world}: backend={backend} device={dev}", flush=True)
    from ctts_amd.synthetic import shard, to_device, as_model_args
    from ctts_amd.trainer import TrainStep
    model, loss_fn, optim = build(block, dev)
    batch = to_device(shard(global_batch(), rank, world), dev)
    step = TrainStep(model, loss_fn, optim, as_model_args(batch), world=world, use_graph=use_graph)
    assert step.staged and step.n_stages == 4
    if use_graph:
        step.capture(warmup=1)       # one eager step (lazy tables, allocator warm-up), then 4 stage graphs + the optimizer graph
    losses = []
    for _ in range(n_steps):
        step()
        losses.append(float(step.loss_val))
    torch.cuda.synchronize()
    torch.save({"params": step.fadam.flat_param.cpu(), "grads": step.arena.flat.cpu(), "losses": losses,
                "norm": float(step.fadam.total_norm), "backend": backend, "device": str(dev)}, out_path)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

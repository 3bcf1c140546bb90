"""One rank of the DistributedDataParallel wrap test (tests/test_dropin_gpu.py; not a test module): the reference's train.py:58
`model = DistributedDataParallel(model, device_ids=[rank])` applied to the product model, whose Conv1d parameters are dense but
STRIDED (GEMM-major memory).  Backend as tests/dp_worker.py picks it: RCCL with `device_ids=[local_rank]` when the box has a GPU per
rank, else gloo with every rank on cuda:0.   argv: out_path
"""
import os
import sys

import torch
import torch.distributed as dist
from torch.nn.parallel import DistributedDataParallel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import dp_worker
    from ctts_amd import ops
    from ctts_amd.synthetic import shard, to_device, as_model_args
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend, dev = dp_worker.pick_backend(rank, world)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    ops.set_grad_accumulation_fusion(False)          # plain autograd: DDP's hooks see every parameter gradient
    model, loss_fn, _ = dp_worker.build("transformer_fs2", dev)
    if rank == 1:                                    # DDP must broadcast rank 0's weights at construction
        with torch.no_grad():
            model.mel_linear.weight.add_(1.0)
    ddp = DistributedDataParallel(model, device_ids=[dev.index])
    args = list(as_model_args(to_device(shard(dp_worker.global_batch(), rank, world), dev)))
    out = ddp(*args, step=50001)
    inputs = [None, None] + args
    inputs[9:11] = out[-2:]
    loss = loss_fn(inputs, out[:-2], 50001)[0]
    loss.backward()
    torch.cuda.synchronize()
    grads = {n: p.grad.detach().cpu().contiguous() for n, p in ddp.module.named_parameters() if p.grad is not None}
    strides_kept = all(p.grad.stride() == p.stride() for p in ddp.module.parameters() if p.grad is not None)
    torch.save({"grads": grads, "loss": float(loss), "strides_kept": strides_kept,
                "mel_w": ddp.module.mel_linear.weight.detach().cpu()}, sys.argv[1])
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

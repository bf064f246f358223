#!/usr/bin/env python3
"""The RCCL-side calls bench.py makes for N > 1, with the one rank a one-GPU box allows (no two ranks may share a device under
RCCL): init_process_group("nccl", device_id=...), the gloo side group, all_reduce / barrier on the RCCL group, all_gather_object
and gather on the gloo group.  Run under torchrun --nproc-per-node 1; the two-rank rehearsals of the whole bench.py path use gloo
(--single-device)."""
import os, time
import torch
import torch.distributed as dist

local_rank = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local_rank)
t0 = time.perf_counter()
dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
host_group = dist.new_group(backend="gloo")
dev = torch.device("cuda", local_rank)
one = torch.ones(1, dtype=torch.float64, device=dev)
dist.all_reduce(one)
devices = [None] * dist.get_world_size()
dist.all_gather_object(devices, local_rank, group=host_group)
flag = torch.tensor([0.0], dtype=torch.float64, device=dev)
dist.all_reduce(flag, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
parts = [torch.empty((4, 3), dtype=torch.int64)] if dist.get_rank() == 0 else None
dist.gather(torch.arange(12, dtype=torch.int64).reshape(4, 3), parts, dst=0, group=host_group)
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
print({"backend": dist.get_backend(), "world": dist.get_world_size(), "ranks_joined": int(one.item()), "devices": devices,
       "max_time": float(t.item()), "gathered": parts[0].sum().item(), "seconds": round(time.perf_counter() - t0, 2)}, flush=True)
dist.destroy_process_group()

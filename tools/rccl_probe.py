#!/usr/bin/env python3
"""Probe: can several RCCL ranks share ONE GPU when every rank claims its own host (NCCL_HOSTID), so that the
duplicate-GPU check passes and the socket transport carries the traffic?  Used to exercise the real RCCL send/recv
code paths on the one-GPU boxes.  Launch under torchrun with --nproc-per-node 2."""
import os
import sys
import time

rank = int(os.environ.get("RANK", "0"))
world = int(os.environ.get("WORLD_SIZE", "1"))
os.environ["NCCL_HOSTID"] = f"smashx-probe-{rank}"
os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
os.environ.setdefault("NCCL_IB_DISABLE", "1")
os.environ.setdefault("NCCL_P2P_DISABLE", "1")
os.environ.setdefault("NCCL_SHM_DISABLE", "1")
import torch
import torch.distributed as dist

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
x = torch.full((1024,), float(rank + 1), device=dev)
dist.all_reduce(x)
torch.cuda.synchronize()
print(f"rank {rank}: all_reduce -> {x[0].item()} (expect {world * (world + 1) / 2})", flush=True)
# point-to-point ring
nxt, prv = (rank + 1) % world, (rank - 1) % world
s = torch.full((4096,), float(rank), device=dev)
r = torch.empty(4096, device=dev)
t0 = time.perf_counter()
for _ in range(20):
    ops = [dist.P2POp(dist.isend, s, nxt), dist.P2POp(dist.irecv, r, prv)]
    for w in dist.batch_isend_irecv(ops):
        w.wait()
torch.cuda.synchronize()
print(f"rank {rank}: p2p got {r[0].item()} (expect {float(prv)}), {1e3 * (time.perf_counter() - t0) / 20:.3f} ms per exchange", flush=True)
dist.barrier()
dist.destroy_process_group()

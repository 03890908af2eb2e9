#!/usr/bin/env python3
"""Achievable HBM rates of this card for the access shapes the kernels have, measured with plain torch ops (plumbing, not product):
streaming write (fill), read (sum), copy (read + write), and four interleaved write streams of 256 B per wave and step like the taped
forward kernel's.  Prints one JSON line; numbers quoted in DESIGN.md section 8."""
import json
import time

import torch

dev = torch.device("cuda", 0)
n = 8 * 1024 ** 3 // 4                      # 8 GiB of fp32 per buffer
a = torch.empty(n, dtype=torch.float32, device=dev)
b = torch.empty(n, dtype=torch.float32, device=dev)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


out = {}
t = timed(lambda: a.fill_(1.0))
out["write_fill_TBps"] = n * 4 / t / 1e12
t = timed(lambda: a.sum())
out["read_sum_TBps"] = n * 4 / t / 1e12
t = timed(lambda: b.copy_(a))
out["copy_read_plus_write_TBps"] = 2 * n * 4 / t / 1e12
# four output streams written from one input (16 B written per 4 B read), row-interleaved like hi / hp / hft / qt of vert_fwd
m = n // 4
src = a[:m]
outs = [torch.empty(m, dtype=torch.float32, device=dev) for _ in range(4)]


def four():
    for o in outs:
        torch.mul(src, 1.0001, out=o)


t = timed(four)
out["four_streams_written_TBps"] = 4 * m * 4 / t / 1e12
out["four_streams_total_TBps"] = 8 * m * 4 / t / 1e12
print(json.dumps({k: round(v, 3) for k, v in out.items()}))

#!/usr/bin/env python3
"""What this box's PCIe link delivers to hipMemcpyAsync from/to page-locked memory: host->device alone, device->host alone,
both at once on two streams (the batch path's situation).  The roof the host_path numbers are read against."""
import json
import time

import torch

n = 1 << 30
h_in = torch.empty(n, dtype=torch.uint8).pin_memory()
h_out = torch.empty(n, dtype=torch.uint8).pin_memory()
d_a = torch.empty(n, dtype=torch.uint8, device="cuda")
d_b = torch.ones(n, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fn, reps=8):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def h2d():
    with torch.cuda.stream(s1):
        d_a.copy_(h_in, non_blocking=True)


def d2h():
    with torch.cuda.stream(s2):
        h_out.copy_(d_b, non_blocking=True)


def both():
    h2d(); d2h()


out = {"bytes_per_copy": n}
out["h2d_GBps"] = round(n / timed(h2d) / 1e9, 2)
out["d2h_GBps"] = round(n / timed(d2h) / 1e9, 2)
t = timed(both)
out["duplex_each_GBps"] = round(n / t / 1e9, 2)
out["duplex_sum_GBps"] = round(2 * n / t / 1e9, 2)
print(json.dumps(out))

#!/usr/bin/env python3
"""Diagnostic: how long does the Skein kernel take while (a) nothing, (b) a plain device memcpy, (c) an ALU-only
torch kernel runs beside it on another stream?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import compute_war_amd as cw

cw.init(0)
bs, nb = 65536, 524288
main = torch.cuda.current_stream()
side = torch.cuda.Stream()
src = torch.empty(nb * bs, dtype=torch.uint8, device="cuda")
cw.dev_gen_random(1, 0, nb, bs, src.data_ptr(), main.cuda_stream)
dig = torch.zeros(nb * 64, dtype=torch.uint8, device="cuda")
a = torch.empty(8 << 30, dtype=torch.uint8, device="cuda")
b = torch.empty(8 << 30, dtype=torch.uint8, device="cuda")
x = torch.randn(64 << 20, device="cuda")
torch.cuda.synchronize()


def run(kind):
    cw.profile_enable(True)
    ev = torch.cuda.Event()
    ev.record(main)
    side.wait_event(ev)
    with torch.cuda.stream(side):
        if kind == "memcpy":
            for _ in range(12):
                b.copy_(a)            # 8 GiB read + 8 GiB write each
        elif kind == "alu":
            y = x
            for _ in range(60):
                y = torch.sin(y) * 1.0001
    cw.dev_hash("skein512", src.data_ptr(), bs, nb, dig.data_ptr(), main.cuda_stream)
    torch.cuda.synchronize()
    p = cw.profile_read()
    print(f"co-runner={kind:7s} skein kernel = {p['hash'][0] / max(p['hash'][1], 1):.2f} ms", flush=True)


for k in ("none", "memcpy", "alu", "none"):
    run(k)

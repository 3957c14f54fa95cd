#!/usr/bin/env python3
"""The fused call (Skein-512 + LZ4 over uniform-random 64 KiB blocks) against its two knobs -- codec scan wavefronts per CU and
number of hash launches per pass -- and against its parts alone: ms per pass, each kernel's own event time, and the effective
shader clock of the pass = GRBM_GUI_ACTIVE / 8 / wall when run under  rocprofv3 --pmc GRBM_GUI_ACTIVE  (MI355X_MICROARCH.md,
"DVFS give-back"; this script only prints the timings, tools/fused_sweep.sh adds the counter).  Not the contract bench.
    python tools/fused_sweep.py [--nb 1048576] [--only fused|hash|scan]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import compute_war_amd as cw  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--nb", type=int, default=1 << 20)
ap.add_argument("--bs", type=int, default=65536)
ap.add_argument("--passes", type=int, default=8)
ap.add_argument("--only", default="")
ap.add_argument("--scan-wpc", default="1,2,4")
ap.add_argument("--slices", default="8,16")
a = ap.parse_args()
cw.init(0)
s = torch.cuda.current_stream().cuda_stream
src = torch.empty(a.nb * a.bs, dtype=torch.uint8, device="cuda")
cw.dev_gen_random(0xC0FFEE, 0, a.nb, a.bs, src.data_ptr(), s)
stride = (cw.compress_bound("lz4", a.bs) + 15) // 16 * 16
dst = torch.empty(a.nb * stride, dtype=torch.uint8, device="cuda")
dig = torch.zeros(a.nb * 64, dtype=torch.uint8, device="cuda")
sizes = torch.zeros(a.nb, dtype=torch.int32, device="cuda")
torch.cuda.synchronize()


def run(name, fn, **knobs):
    with cw.tuned(**knobs):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        cw.profile_enable(True)
        cw.profile_read(reset=True)
        t0 = time.perf_counter()
        for _ in range(a.passes):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.passes * 1e3
        p = cw.profile_read(reset=True)
        cw.profile_enable(False)
    rec = {"config": name, "knobs": knobs, "ms_per_pass": round(dt, 2), "GBps": round(a.nb * a.bs / dt / 1e6, 1),
           "hash_ms": round(p["hash"][0] / max(p["hash"][1], 1), 2), "codec_ms": round(p["codec"][0] / max(p["codec"][1], 1), 2)}
    print(json.dumps(rec), flush=True)


def hash_only():
    cw.dev_hash("skein512", src.data_ptr(), a.bs, a.nb, dig.data_ptr(), s)


def scan_only():
    cw.dev_compress("lz4", src.data_ptr(), a.bs, a.nb, dst.data_ptr(), stride, sizes.data_ptr(), s)


def fused():
    cw.dev_hash_and_compress("skein512", "lz4", src.data_ptr(), a.bs, a.nb, dig.data_ptr(), dst.data_ptr(), stride, sizes.data_ptr(), s)


if a.only in ("", "hash"):
    run("hash alone", hash_only)
if a.only in ("", "scan"):
    run("scan alone", scan_only)
if a.only in ("", "fused"):
    run("fused, shipped (4 scan wavefronts per CU, 8 hash launches)", fused)
    for w in [int(x) for x in a.scan_wpc.split(",")]:
        for k in [int(x) for x in a.slices.split(",")]:
            run(f"fused, scan wpc {w}, {k} hash launches", fused, CW_SCAN_WPC=w, CW_SKEIN_NSLICES=k)
    run("fused, both kernels on one stream (CW_SERIAL=1)", fused, CW_SERIAL=1)

#!/usr/bin/env python3
"""In-kernel clock of the hash kernel alone, of the LZ4 span scan alone and of both side by side (the fused call), from the
diagnostic build with clock stamps (make -C compute_war_amd/csrc clock -> libcwhc_clock.so):
    clock = d(s_memtime) / d(s_memrealtime) x 100 MHz per workgroup, median over the workgroups of the last launches,
after >= 2 s of back-to-back launches on random data (MI355X_MICROARCH.md, "DVFS give-back" item 6).  This replaces the
rocm-smi reading of round 1 as the evidence for "the fused call runs at a lower clock than either kernel alone".
    CW_LIB=compute_war_amd/libcwhc_clock.so python tools/clock_probe.py [--nb 1048576]
"""
import argparse
import ctypes as C
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CW_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "compute_war_amd", "libcwhc_clock.so"))
import torch  # noqa: E402

import compute_war_amd as cw  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--nb", type=int, default=1 << 20)
ap.add_argument("--bs", type=int, default=65536)
ap.add_argument("--seconds", type=float, default=2.5)
a = ap.parse_args()
cw.init(0)
L = cw.lib()
s = torch.cuda.current_stream().cuda_stream
src = torch.empty(a.nb * a.bs, dtype=torch.uint8, device="cuda")
cw.dev_gen_random(0xC0FFEE, 0, a.nb, a.bs, src.data_ptr(), s)
stride = (cw.compress_bound("lz4", a.bs) + 15) // 16 * 16
dst = torch.empty(a.nb * stride, dtype=torch.uint8, device="cuda")
dig = torch.zeros(a.nb * 64, dtype=torch.uint8, device="cuda")
sizes = torch.zeros(a.nb, dtype=torch.int32, device="cuda")
torch.cuda.synchronize()


def clocks(which, lo=0, hi=1024):
    buf = (C.c_ulonglong * 4096)()
    L.cw_debug_clock_read.argtypes = [C.c_int, C.c_void_p]
    assert L.cw_debug_clock_read(which, buf) == 0
    out = []
    for i in range(lo, hi):
        m0, r0, m1, r1 = buf[4 * i:4 * i + 4]
        if r1 > r0 and m1 > m0 and (r1 - r0) > 1000:   # > 10 us of 100 MHz ticks
            out.append((m1 - m0) / (r1 - r0) * 100e6 / 1e9)
    return out


def run(name, fn):
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < a.seconds:
        fn()
        n += 1
        if n % 4 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res = {"passes": n, "ms_per_pass": round(dt / n * 1e3, 2)}
    for which, k in ((0, "skein_slice_kernel"), (1, "lz4_scan_span_kernel")):
        c = clocks(which)
        if c:
            res[k] = {"GHz_median": round(statistics.median(c), 3), "GHz_min": round(min(c), 3), "GHz_max": round(max(c), 3), "workgroups": len(c)}
    # the hash runs as 8 launches per pass; the stamps are kept per launch index (128 workgroups each): in the fused call the first
    # launches overlap the codec's scan, the last ones run after it has ended
    per = []
    for k in range(8):
        c = clocks(0, 128 * k, 128 * (k + 1))
        per.append(round(statistics.median(c), 3) if c else None)
    res["skein_slice_kernel_by_launch_GHz"] = per
    return res


def zero_stamps():
    pass  # stamps of earlier modes are overwritten slot by slot; modes that do not launch a kernel are filtered below


out = {"method": "d(s_memtime)/d(s_memrealtime) x 100 MHz per workgroup around the kernel body, median over <= 1024 workgroups of the last launches; "
                 f">= {a.seconds} s of back-to-back launches on uniform-random data, {a.nb} x {a.bs} B blocks",
       "lib": os.path.basename(cw.lib_path())}
out["hash_alone"] = run("hash", lambda: cw.dev_hash("skein512", src.data_ptr(), a.bs, a.nb, dig.data_ptr(), s))
out["hash_alone"].pop("lz4_scan_span_kernel", None)
out["scan_alone"] = run("scan", lambda: cw.dev_compress("lz4", src.data_ptr(), a.bs, a.nb, dst.data_ptr(), stride, sizes.data_ptr(), s))
out["scan_alone"].pop("skein_slice_kernel", None)
out["scan_alone"].pop("skein_slice_kernel_by_launch_GHz", None)   # (stale stamps of the previous mode)
out["fused"] = run("fused", lambda: cw.dev_hash_and_compress("skein512", "lz4", src.data_ptr(), a.bs, a.nb, dig.data_ptr(), dst.data_ptr(), stride,
                                                             sizes.data_ptr(), s))
print(json.dumps(out, indent=1))

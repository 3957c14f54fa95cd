#!/usr/bin/env python3
"""The host batch path alone (cw_hash_and_compress_packed over pinned buffers), for timeline traces:
    rocprofv3 --kernel-trace --memory-copy-trace -d out -o p -f csv -- python3 tools/host_path_probe.py [--gib 8]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import compute_war_amd as cw  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--gib", type=float, default=8.0)
ap.add_argument("--bs", type=int, default=65536)
ap.add_argument("--hash", default="skein512")
ap.add_argument("--comp", default="lz4")
ap.add_argument("--passes", type=int, default=2)
ap.add_argument("--data", default="random", choices=["random", "corpus"])
ap.add_argument("--warm", action="store_true", help="one device->host copy over the output buffer before the first pass")
a = ap.parse_args()
cw.init(0)
L = cw.lib()
H = {"skein512": 0, "skein": 1, "sha256mb": 2}[a.hash]
Cc = {"lz4": 0, "lzf": 1}[a.comp]
nb = int(a.gib * (1 << 30)) // a.bs
cap = nb * cw.compress_bound(a.comp, a.bs)
hs, hp = L.cw_host_alloc(nb * a.bs), L.cw_host_alloc(cap)
dev = torch.empty(nb * a.bs, dtype=torch.uint8, device="cuda")
if a.data == "random":
    cw.dev_gen_random(0xC0FFEE, 0, nb, a.bs, dev.data_ptr(), torch.cuda.current_stream().cuda_stream)
else:  # the bench legs' corpus tile (canterbury + canterbury-large, whole 64 KiB blocks of every file), tiled
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    tile = torch.frombuffer(bytearray(bench.corpus_tile()[0]), dtype=torch.uint8).cuda()
    for o in range(0, nb * a.bs, tile.numel()):
        k = min(tile.numel(), nb * a.bs - o)
        dev[o:o + k] = tile[:k]
torch.cuda.synchronize()
cw.ops.check(L.cw_dev_download(hs, dev.data_ptr(), nb * a.bs))
del dev
torch.cuda.empty_cache()
dig = np.zeros((nb, cw.digest_bytes(a.hash)), dtype=np.uint8)
sizes = np.zeros(nb, dtype=np.uint32)
offs = np.zeros(nb + 1, dtype=np.uint64)
cw.ops.check(L.cw_prepare(H, Cc, a.bs, nb, 1))
if a.warm:
    t0 = time.perf_counter()
    scratch = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
    for o in range(0, cap, 1 << 30):
        cw.ops.check(L.cw_dev_download(hp + o, scratch.data_ptr(), min(1 << 30, cap - o)))
    del scratch
    print(f"warm-up copy over the output buffer: {(time.perf_counter() - t0) * 1e3:.0f} ms", flush=True)
for i in range(a.passes):
    cw.profile_enable(True)
    cw.profile_read(reset=True)
    t0 = time.perf_counter()
    cw.ops.check(L.cw_hash_and_compress_packed(H, Cc, hs, a.bs, nb, dig.ctypes.data, hp, cap, offs.ctypes.data, sizes.ctypes.data))
    t = time.perf_counter() - t0
    print(f"pass {i}: {nb * a.bs / t / 1e9:.2f} GB/s ({t * 1e3:.1f} ms), out {int(offs[nb])} bytes", flush=True)

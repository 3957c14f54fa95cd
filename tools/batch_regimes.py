#!/usr/bin/env python3
"""Throughput of the codecs (alone and beside Skein-512) against the NUMBER of blocks in a call (not the contract bench).

The parse kernels are latency-bound chains, one per block, so what a call reaches depends on how many blocks it brings:
below ~2,560 blocks of 64 KiB the wavefront-per-block parser has free slots, up to ~24 Ki blocks it works in rounds of
2,560, from there on the lane-per-block parsers take over.  BASELINE configs[2] as literally stated
(Silesia, 211 MB = 3,233 blocks of 64 KiB) sits in the second regime; the bench legs tile the corpus to 4 GiB.
  gpurun -- 'python tools/batch_regimes.py > gpurun_out/batch_regimes.txt'
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import compute_war_amd as cw  # noqa: E402

import argparse  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--sizes", default="64,256,1024,2560,3233,5120,8192,16384,24576,32768,65536,262144", help="numbers of 64 KiB blocks (x16 for 4 KiB)")
ap.add_argument("--bs", default="65536,4096")
ap.add_argument("--tag", default="")
A = ap.parse_args()
cw.init(0)
s = torch.cuda.current_stream().cuda_stream
text = open(os.path.join(os.path.dirname(__file__), "..", "tests/golden/corpus/canterbury/lcet10.txt"), "rb").read()


def rate(fn, nbytes, iters):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return nbytes * iters / (time.perf_counter() - t0) / 1e9


print(f"{'blocks':>8s} {'MiB':>7s} | {'lz4':>7s} {'+skein512':>9s} | {'lzf':>7s} {'+sha256':>8s}   (GB/s; lcet10.txt tiled, 64 KiB blocks)")
if A.tag:
    print("#", A.tag, {k: v for k, v in os.environ.items() if k.startswith("CW_")})
for bs in [int(x) for x in A.bs.split(",")]:
    if bs == 4096:
        print(f"{'blocks':>8s} {'MiB':>7s} | {'lz4':>7s} {'+skein256':>9s} | {'lzf':>7s} {'+sha256':>8s}   (GB/s; 4 KiB blocks)")
    for nb64 in [int(x) for x in A.sizes.split(",")]:
        nb = nb64 * (65536 // bs)
        src = torch.empty(nb * bs, dtype=torch.uint8, device="cuda")
        t = (text * (nb * bs // len(text) + 1))[: nb * bs]
        src.copy_(torch.frombuffer(bytearray(t), dtype=torch.uint8))
        stride = (cw.compress_bound("lz4", bs) + 15) // 16 * 16
        dst = torch.empty(nb * stride, dtype=torch.uint8, device="cuda")
        sizes = torch.zeros(nb, dtype=torch.int32, device="cuda")
        dig = torch.zeros(nb * 64, dtype=torch.uint8, device="cuda")
        iters = 3 if nb64 >= 16384 else 10
        row = []
        for comp, h in (("lz4", "skein512" if bs == 65536 else "skein"), ("lzf", "sha256mb")):
            row.append(rate(lambda: cw.dev_compress(comp, src.data_ptr(), bs, nb, dst.data_ptr(), stride, sizes.data_ptr(), s), nb * bs, iters))
            row.append(rate(lambda: cw.dev_hash_and_compress(h, comp, src.data_ptr(), bs, nb, dig.data_ptr(), dst.data_ptr(), stride,
                                                             sizes.data_ptr(), s), nb * bs, iters))
        print(f"{nb:8d} {nb * bs / 2**20:7.0f} | {row[0]:7.2f} {row[1]:9.2f} | {row[2]:7.2f} {row[3]:8.2f}", flush=True)
        del src, dst, sizes, dig
        torch.cuda.empty_cache()

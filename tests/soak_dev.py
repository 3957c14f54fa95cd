#!/usr/bin/env python3
"""Time-boxed randomized soak of the DEVICE entry points with strides and misaligned bases (test infrastructure; run by hand on
a GPU box):  python tests/soak_dev.py 120 [seed]   (CW_LZ4_LANES=1 CW_LZF_LANES=1 CW_LZF_ROUND=7 CW_DECODE_LANES=1: the lane kernels)
cw_dev_compress / cw_dev_hash_and_compress / cw_dev_decompress with a source stride larger than the block, slot strides that
are not multiples of anything, source and slot bases at odd addresses, buffers that END where the last block / slot ends --
every size, payload byte and digest against the oracle, every slot decoded back."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import compute_war_amd as cw  # noqa: E402
import oracle  # noqa: E402
from test_gpu_fuzz import _block  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
cw.init(0)
s = torch.cuda.current_stream().cuda_stream
HASH = {"skein512": lambda b: oracle.skein512(b, 512), "skein": lambda b: oracle.skein256(b, 128), "sha256mb": oracle.sha256}
SIZES = [4096, 65536, 2048, 8192, 16384, 32768, 1000, 333, 65535, 4097, 4095, 64, 20000, 48, 16, 13]
t0, rounds, blocks, bad = time.time(), 0, 0, 0
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed0 + rounds)
    n = int(rng.choice(SIZES)) if rng.random() < 0.8 else int(rng.integers(1, 65537))
    count = int(rng.integers(1, max(2, min(300, (4 << 20) // n))))
    calg = ("lz4", "lzf")[rounds % 2]
    halg = [None, "skein512", "skein", "sha256mb"][(rounds // 2) % 4]
    sstride = n + int(rng.choice([0, 0, 1, 7, 16, 48]))
    dstride = cw.compress_bound(calg, n) + int(rng.choice([0, 3, 13, 16, 29]))
    sshift, dshift = int(rng.choice([0, 0, 1, 2, 3, 8, 16])), int(rng.choice([0, 0, 1, 5, 16]))
    if os.environ.get("CW_SOAK_VERBOSE"):
        print("round", rounds, calg, halg, "n", n, "count", count, "sstride", sstride, "dstride", dstride, "shifts", sshift, dshift, flush=True)
    data = [_block(rng, n) if n >= 16 else rng.integers(0, 3, n, dtype=np.uint8) for _ in range(count)]
    src = np.zeros(sshift + sstride * (count - 1) + n, dtype=np.uint8)            # ends with the last block's last byte
    for i in range(count):
        src[sshift + i * sstride: sshift + i * sstride + n] = data[i]
    d_src = torch.from_numpy(src).cuda()
    d_dst = torch.zeros(dshift + dstride * count, dtype=torch.uint8, device="cuda")  # ends with the last slot
    d_sz = torch.zeros(count, dtype=torch.int32, device="cuda")
    db = cw.digest_bytes(halg) if halg else 0
    d_dig = torch.zeros(count * db + 1, dtype=torch.uint8, device="cuda")
    if halg:
        cw.dev_hash_and_compress(halg, calg, d_src.data_ptr() + sshift, n, count, d_dig.data_ptr(), d_dst.data_ptr() + dshift, dstride, d_sz.data_ptr(), s,
                                 src_stride=sstride)
    else:
        cw.dev_compress(calg, d_src.data_ptr() + sshift, n, count, d_dst.data_ptr() + dshift, dstride, d_sz.data_ptr(), s, src_stride=sstride)
    d_back = torch.zeros(count * n, dtype=torch.uint8, device="cuda")
    d_st = torch.full((count,), 7, dtype=torch.int32, device="cuda")
    cw.dev_decompress(calg, d_dst.data_ptr() + dshift, dstride, d_sz.data_ptr(), count, d_back.data_ptr(), n, d_st.data_ptr(), s)
    torch.cuda.synchronize()
    hsz, hdst, hdig = d_sz.cpu().numpy(), d_dst.cpu().numpy(), d_dig.cpu().numpy()
    hback, hst = d_back.cpu().numpy().reshape(count, n), d_st.cpu().numpy()
    enc = oracle.lz4_compress if calg == "lz4" else oracle.lzf_compress
    for i in range(count):
        blk = data[i].tobytes()
        want = enc(blk)
        o = dshift + i * dstride
        ok = int(hsz[i]) == len(want) and hdst[o:o + len(want)].tobytes() == want
        if halg:
            ok = ok and hdig[i * db:(i + 1) * db].tobytes() == HASH[halg](blk)
        if len(want):
            ok = ok and int(hst[i]) == 0 and hback[i].tobytes() == blk
        else:
            ok = ok and int(hst[i]) == 1
        if not ok:
            bad += 1
            print("MISMATCH", calg, halg, "n", n, "block", i, "of", count, "sstride", sstride, "dstride", dstride, "shifts", sshift, dshift, "seed", seed0 + rounds, flush=True)
    rounds += 1
    blocks += count
print(f"soak: {rounds} rounds, {blocks} blocks (size, payload, digest, decode) against the oracle in {time.time() - t0:.0f} s, seed0 {seed0}: {bad} mismatches", flush=True)
sys.exit(1 if bad else 0)

#!/usr/bin/env python3
"""Time-boxed randomized soak of the hash kernels against the oracle (test infrastructure; run by hand on a GPU box):
    python tests/soak_hash.py 120 [seed]
Random block sizes (0 .. 65536, biased to the sizes with their own kernel variants), block counts on both sides of the kernel
selection thresholds, all three algorithms, host-batch and device entry points (contiguous and strided), the fused call."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import compute_war_amd as cw  # noqa: E402
import oracle  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
cw.init(0)
s = torch.cuda.current_stream().cuda_stream
REF = {"skein512": lambda b: oracle.skein512(b, 512), "skein": lambda b: oracle.skein256(b, 128), "sha256mb": oracle.sha256}
SIZES = [4096, 65536, 64, 128, 32, 31, 33, 63, 65, 1, 0, 100, 1000, 4095, 4097, 8192, 16384, 32768, 65535, 12345]
t0, rounds, digests, bad = time.time(), 0, 0, 0
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed0 + rounds)
    n = int(rng.choice(SIZES)) if rng.random() < 0.7 else int(rng.integers(0, 65537))
    count = int(rng.choice([1, 2, 63, 64, 65, 200, 4096, 4100])) if n <= 4096 and rng.random() < 0.5 else int(rng.integers(1, max(2, min(300, (4 << 20) // max(n, 1)))))
    alg = ["skein512", "skein", "sha256mb"][rounds % 3]
    data = rng.integers(0, 256, max(n * count, 1), dtype=np.uint8)
    check = sorted(set([0, count - 1] + [int(x) for x in rng.integers(0, count, min(count, 24))]))
    want = {i: REF[alg](data[i * n:(i + 1) * n].tobytes()) for i in check}
    mode = rounds % 4
    if os.environ.get("CW_SOAK_VERBOSE"):
        print("round", rounds, alg, "n", n, "count", count, "mode", mode, flush=True)
    if mode == 0 and n > 0:      # host batch entry point
        got = cw.hash_blocks(alg, data[:n * count], n)
    else:                        # device entry point, strided and misaligned by `shift`
        stride = n + int(rng.choice([0, 0, 16, 7]))
        shift = int(rng.choice([0, 0, 1, 3, 8]))
        buf = np.zeros(stride * count + shift + 64, dtype=np.uint8)
        for i in range(count):
            buf[shift + i * stride: shift + i * stride + n] = data[i * n:(i + 1) * n]
        d = torch.from_numpy(buf).cuda()
        db = cw.digest_bytes(alg)
        dig = torch.zeros(count * db + 16, dtype=torch.uint8, device="cuda")
        if mode == 3 and n > 0:  # beside a codec (the fused call picks other hash kernel variants)
            cs = (cw.compress_bound("lz4", n) + 15) // 16 * 16
            dst = torch.zeros(count * cs, dtype=torch.uint8, device="cuda")
            sz = torch.zeros(count, dtype=torch.int32, device="cuda")
            cw.dev_hash_and_compress(alg, "lz4", d.data_ptr() + shift, n, count, dig.data_ptr(), dst.data_ptr(), cs, sz.data_ptr(), s, src_stride=stride)
        else:
            cw.dev_hash(alg, d.data_ptr() + shift, n, count, dig.data_ptr(), s, src_stride=stride)
        torch.cuda.synchronize()
        got = dig[:count * db].cpu().numpy().reshape(count, db)
    for i in check:
        if got[i].tobytes() != want[i]:
            bad += 1
            print("MISMATCH", alg, "n", n, "block", i, "of", count, "mode", mode, "seed", seed0 + rounds, flush=True)
    rounds += 1
    digests += len(check)
print(f"soak: {rounds} rounds, {digests} digests checked against the oracle in {time.time() - t0:.0f} s, seed0 {seed0}: {bad} mismatches", flush=True)
sys.exit(1 if bad else 0)

#!/usr/bin/env python3
"""Time-boxed randomized soak of the device decoders on MALFORMED input (test infrastructure; run by hand on a GPU box):
    python tests/soak_decode.py 120 [seed]                      # wavefront decoders
    CW_DECODE_LANES=1 python tests/soak_decode.py 120 [seed]    # lane-per-block decoders
Valid LZ4 / LZF streams of synthetic blocks, then per slot: bytes flipped, the size cut or stretched over garbage, the whole slot
garbage, a size beyond the slot, size 0.  Status and (when a stream still decodes to exactly one block) output must equal the
oracle decoder's verdict; nothing may fault."""
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
t0, rounds, slots, bad, still_ok = time.time(), 0, 0, 0, 0
SIZES = [4096, 65536, 2048, 1024, 8192, 16384, 32768, 20000, 333, 100, 65535, 4097, 64, 17]
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed0 + rounds)
    n = int(rng.choice(SIZES))
    count = int(rng.integers(1, max(2, min(200, (3 << 20) // n))))
    alg = ("lz4", "lzf")[rounds % 2]
    enc, dec = (oracle.lz4_compress, oracle.lz4_decompress) if alg == "lz4" else (oracle.lzf_compress, oracle.lzf_decompress)
    stride = (cw.compress_bound(alg, n) + 15) // 16 * 16
    buf = rng.integers(0, 256, (count, stride), dtype=np.uint8)       # garbage behind every stream
    sz = np.zeros(count, dtype=np.int64)
    for i in range(count):
        c = enc(_block(rng, n).tobytes()) if n >= 16 else enc(rng.integers(0, 3, n, dtype=np.uint8).tobytes())
        buf[i, :len(c)] = np.frombuffer(c, dtype=np.uint8)
        sz[i] = len(c)
        m = rng.random()
        if m < 0.25 or sz[i] == 0:
            pass                                                       # untouched (LZF: 0 = did not fit)
        elif m < 0.5:
            for _ in range(int(rng.integers(1, 4))):
                buf[i, int(rng.integers(0, sz[i]))] = rng.integers(0, 256)
        elif m < 0.65:
            sz[i] = int(rng.integers(1, sz[i] + 1))                    # cut
        elif m < 0.8:
            sz[i] = int(rng.integers(sz[i], stride + 1))               # stretched over the garbage behind the stream
        elif m < 0.9:
            buf[i] = rng.integers(0, 256, stride, dtype=np.uint8)
            sz[i] = int(rng.integers(1, stride + 1))
        else:
            sz[i] = int(rng.choice([0, stride + 1, stride + 1000, 0xFFFFFFF0, 1 << 25]))
    d_buf = torch.from_numpy(buf.reshape(-1)).cuda()
    d_sz = torch.from_numpy(sz.astype(np.uint32).view(np.int32)).cuda()
    out = torch.zeros(count * n, dtype=torch.uint8, device="cuda")
    st = torch.full((count,), 7, dtype=torch.int32, device="cuda")
    cw.dev_decompress(alg, d_buf.data_ptr(), stride, d_sz.data_ptr(), count, out.data_ptr(), n, st.data_ptr(), s)
    torch.cuda.synchronize()
    hst, hout = st.cpu().numpy(), out.cpu().numpy().reshape(count, n)
    for i in range(count):
        r = dec(buf[i, :sz[i]].tobytes(), n) if 0 < sz[i] <= stride else None
        ok = r is not None and len(r) == n
        still_ok += ok
        if int(hst[i]) != (0 if ok else 1) or (ok and hout[i].tobytes() != r):
            bad += 1
            print("MISMATCH", alg, "n", n, "slot", i, "of", count, "size", int(sz[i]), "status", int(hst[i]), "oracle ok", ok, "seed", seed0 + rounds, flush=True)
    rounds += 1
    slots += count
print(f"soak: {rounds} rounds, {slots} slots ({still_ok} of them still valid streams) decoded on the device and by the oracle in {time.time() - t0:.0f} s, "
      f"seed0 {seed0}: {bad} disagreements", flush=True)
sys.exit(1 if bad else 0)

#!/usr/bin/env python3
"""Soak of the launch policy's own regimes: for a list of (codec, block size, batch size) the sizes and a Skein-512 digest per
64 KiB of the packed output stream, printed one line each.  Run twice -- as is, and with CW_LZ4_LANES=0 CW_LZF_LANES=0 (the
wavefront parsers alone, which the parity tests pin to the oracle) -- and diff the outputs:
    python tools/soak_side_by_side.py > a.txt; CW_LZ4_LANES=0 CW_LZF_LANES=0 python tools/soak_side_by_side.py > b.txt; diff a.txt b.txt
Data: the in-tree corpora tiled, with a 64 KiB run of noise every 5 x 64 KiB and a different rotation per case."""
import glob
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import compute_war_amd as cw  # noqa: E402

cw.init(0)
s = torch.cuda.current_stream().cuda_stream
data = b"".join(open(f, "rb").read() for f in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "corpus", "*", "*"))))
rng = np.random.default_rng(2026)
CASES = [("lzf", 4096, nb) for nb in (49152, 57344, 65536, 70001, 98304, 131072, 200000, 524288)] + \
        [("lzf", 2048, nb) for nb in (65536, 150000)] + [("lzf", 1000, 120000)] + \
        [("lz4", 4096, nb) for nb in (98304, 100001, 131072, 262144, 524288)] + [("lz4", 2048, 200000), ("lz4", 1000, 150000)] + \
        [("lzf", 65536, nb) for nb in (24576, 30000, 53248, 57344, 65536, 81920, 100001)] + \
        [("lz4", 65536, nb) for nb in (22528, 24576, 30000, 40960, 57344, 65536, 81920, 98304, 100001, 140000)] + \
        [("lzf", 16384, 100000), ("lz4", 16384, 100000), ("lzf", 20000, 30000), ("lz4", 20000, 30000)]
for ci, (comp, bs, nb) in enumerate(CASES):
    total = nb * bs
    rot = (ci * 7919 * 4096) % len(data)
    a = np.frombuffer(((data[rot:] + data[:rot]) * (total // len(data) + 1))[:total], dtype=np.uint8).copy()
    for o in range(ci * 65536, total - 65536, 5 * 65536):
        a[o:o + 65536] = rng.integers(0, 256, 65536, dtype=np.uint8)
    src = torch.from_numpy(a).cuda()
    del a
    stride = (cw.compress_bound(comp, bs) + 15) // 16 * 16
    dst = torch.zeros(nb * stride, dtype=torch.uint8, device="cuda")
    sizes = torch.zeros(nb, dtype=torch.int32, device="cuda")
    lines = []
    for rep in range(3):
        sizes.fill_(-7)
        cw.dev_compress(comp, src.data_ptr(), bs, nb, dst.data_ptr(), stride, sizes.data_ptr(), s)
        offs = torch.zeros(nb + 1, dtype=torch.int64, device="cuda")
        cw.dev_pack(dst.data_ptr(), stride, sizes.data_ptr(), nb, 0, offs.data_ptr(), s)
        torch.cuda.synchronize()
        tot = int(offs[-1].item())
        nd = (tot + 65535) // 65536
        packed = torch.zeros(nd * 65536, dtype=torch.uint8, device="cuda")
        cw.dev_pack(dst.data_ptr(), stride, sizes.data_ptr(), nb, packed.data_ptr(), offs.data_ptr(), s)
        dig = torch.zeros(nd * 64, dtype=torch.uint8, device="cuda")
        cw.dev_hash("skein512", packed.data_ptr(), 65536, nd, dig.data_ptr(), s)
        torch.cuda.synchronize()
        h = hashlib.sha256(sizes.cpu().numpy().tobytes())
        h.update(dig.cpu().numpy().tobytes())
        lines.append(f"{comp} {bs} {nb} out={tot} raw={int((sizes == 0).sum().item())} untouched={int((sizes == -7).sum().item())} {h.hexdigest()}")
        del packed, dig, offs
    assert lines[0] == lines[1] == lines[2], lines   # the same call three times: the same bytes
    print(lines[0], flush=True)
    print("#", cw.profile_kernels()["codec"], file=sys.stderr, flush=True)
    del src, dst, sizes
    torch.cuda.empty_cache()

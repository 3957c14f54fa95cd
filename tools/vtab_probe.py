#!/usr/bin/env python3
"""LZ4 (or, with --comp lzf, LZF) parsers against each other on one input: throughput per launch-policy setting and, for every setting, every block's size and
payload digest against the oracle (not the contract bench; a profiling tool).

  gpurun -- 'python tools/vtab_probe.py --sizes 1024,3233,16384 > gpurun_out/vtab_probe.txt'
Settings are knob sets given to cw_tune_set (in-process), e.g. "CW_LZ4_VTAB=1" or "CW_LZ4_VTAB=2,CW_VTAB_RESERVE=512"."""
import argparse
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import compute_war_amd as cw  # noqa: E402
import oracle as O  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--sizes", default="256,1024,3233,8192,16384", help="numbers of 64 KiB blocks (x16 for 4 KiB)")
ap.add_argument("--bs", default="65536")
ap.add_argument("--data", default="text", choices=["text", "corpus"])
ap.add_argument("--settings", default="default;CW_LZ4_VTAB=1;CW_LZ4_VTAB=2")
ap.add_argument("--no-check", action="store_true")
ap.add_argument("--comp", default="lz4", choices=["lz4", "lzf"])
A = ap.parse_args()
cw.init(0)
O.build()
s = torch.cuda.current_stream().cuda_stream
if A.data == "text":
    base = open(os.path.join(ROOT, "tests/golden/corpus/canterbury/lcet10.txt"), "rb").read()
    base = base[: len(base) // 65536 * 65536]
else:
    from conftest import corpus_file, corpus_large_file, corpus_large_names, corpus_names
    base = b"".join(corpus_file(n)[: len(corpus_file(n)) // 65536 * 65536] for n in corpus_names())
    base += b"".join(corpus_large_file(n)[: len(corpus_large_file(n)) // 65536 * 65536] for n in corpus_large_names())


def rate(fn, nbytes, iters):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return nbytes * iters / (time.perf_counter() - t0) / 1e9


settings = [x for x in A.settings.split(";") if x]
for bs in [int(x) for x in A.bs.split(",")]:
    T = len(base) // bs
    _, _, osz, opay = O.hash_and_compress(np.frombuffer(base, dtype=np.uint8), bs, O.HASH_NONE, O.COMP_LZ4 if A.comp == "lz4" else O.COMP_LZF, 8, want_payload=True)
    odig = [hashlib.blake2b(opay[i, : int(osz[i])].tobytes(), digest_size=8).digest() for i in range(T)]
    print(f"# {A.data}: {T} x {bs} B blocks per period, ratio {T * bs / float(osz.sum()):.4f}", flush=True)
    for nb64 in [int(x) for x in A.sizes.split(",")]:
        nb = nb64 * (65536 // bs)
        t = (base * (nb * bs // len(base) + 1))[: nb * bs]
        src = torch.frombuffer(bytearray(t), dtype=torch.uint8).cuda()
        stride = (cw.compress_bound(A.comp, bs) + 15) // 16 * 16
        dst = torch.zeros(nb * stride, dtype=torch.uint8, device="cuda")
        sizes = torch.zeros(nb, dtype=torch.int32, device="cuda")
        want_sz = torch.from_numpy(osz.astype(np.int32)).cuda().repeat((nb + T - 1) // T)[:nb]
        for st in settings:
            knobs = {} if st == "default" else dict(kv.split("=") for kv in st.split(","))
            with cw.tuned(**knobs):
                dst.zero_()
                sizes.zero_()
                r = rate(lambda: cw.dev_compress(A.comp, src.data_ptr(), bs, nb, dst.data_ptr(), stride, sizes.data_ptr(), s), nb * bs,
                         3 if nb64 >= 16384 else 8)
                names = cw.profile_kernels()["codec"]
            verdict = "unchecked"
            if not A.no_check:
                bad = int((sizes != want_sz).sum().item())
                verdict = f"sizes {nb - bad}/{nb}"
                if bad == 0:   # payload of up to two periods + the last blocks, byte for byte through a digest per block
                    pick = list(range(min(nb, 2 * T))) + list(range(max(0, nb - T), nb))
                    slots = dst.view(nb, stride)
                    wrong = 0
                    for i in pick:
                        z = int(osz[i % T])
                        if hashlib.blake2b(slots[i, :z].cpu().numpy().tobytes(), digest_size=8).digest() != odig[i % T]:
                            wrong += 1
                    verdict += f", payload {len(pick) - wrong}/{len(pick)}"
                else:
                    i = int(torch.nonzero(sizes != want_sz)[0].item())
                    verdict += f" FIRST BAD block {i}: {int(sizes[i])} != {int(want_sz[i])}"
            print(f"{nb:8d} x {bs:5d}  {r:8.2f} GB/s  {st:40s} {verdict:40s} {names}", flush=True)
        del src, dst, sizes, want_sz
        torch.cuda.empty_cache()

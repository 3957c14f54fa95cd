#!/usr/bin/env python3
"""Quick device-side throughput probe of one hash / codec kernel (not the contract bench)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import compute_war_amd as cw  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--alg", default="skein512")
ap.add_argument("--comp", default="")
ap.add_argument("--bs", type=int, default=65536)
ap.add_argument("--nb", type=int, default=262144)
ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--tree", default="", help="leaf,node,maxlevel: Skein tree hashing instead of sequential")
ap.add_argument("--data", default="random", choices=["random", "zero", "text", "mixed"])
a = ap.parse_args()

cw.init(0)
s = torch.cuda.current_stream().cuda_stream
src = torch.empty(a.nb * a.bs, dtype=torch.uint8, device="cuda")
if a.data == "random":
    cw.dev_gen_random(0xC0FFEE, 0, a.nb, a.bs, src.data_ptr(), s)
elif a.data == "mixed":
    cw.dev_gen_mixed(0xC0FFEE, 0, a.nb, a.bs, src.data_ptr(), s)
elif a.data == "zero":
    src.zero_()
else:
    t = open(os.path.join(os.path.dirname(__file__), "..", "tests/golden/corpus/canterbury/lcet10.txt"), "rb").read()
    t = (t * (a.nb * a.bs // len(t) + 1))[: a.nb * a.bs]
    src.copy_(torch.frombuffer(bytearray(t), dtype=torch.uint8))
dig = torch.zeros(a.nb * 64, dtype=torch.uint8, device="cuda")
if a.comp:
    stride = (cw.compress_bound(a.comp, a.bs) + 15) // 16 * 16
    dst = torch.empty(a.nb * stride, dtype=torch.uint8, device="cuda")
    sizes = torch.zeros(a.nb, dtype=torch.int32, device="cuda")


def run():
    if a.comp and a.alg != "none":
        cw.dev_hash_and_compress(a.alg, a.comp, src.data_ptr(), a.bs, a.nb, dig.data_ptr(), dst.data_ptr(), stride, sizes.data_ptr(), s)
    elif a.comp:
        cw.dev_compress(a.comp, src.data_ptr(), a.bs, a.nb, dst.data_ptr(), stride, sizes.data_ptr(), s)
    elif a.tree:
        leaf, node, ml = (int(x) for x in a.tree.split(","))
        cw.dev_hash_tree(a.alg, src.data_ptr(), a.bs, a.nb, leaf, node, ml, dig.data_ptr(), s)
    else:
        cw.dev_hash(a.alg, src.data_ptr(), a.bs, a.nb, dig.data_ptr(), s)


run()
torch.cuda.synchronize()
cw.profile_enable(True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.iters):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.iters
gb = a.nb * a.bs / 1e9
extra = ""
if a.comp:
    marked = int((sizes == -1).sum().item())
    tot = float(sizes[sizes != -1].sum().item() + marked * a.bs)
    extra = f" ratio={a.nb * a.bs / max(tot, 1.0):.4f} marked={marked}"
prof = cw.profile_read()
extra += " | kernel ms: " + ", ".join(f"{k}={v[0] / max(v[1], 1):.2f}" for k, v in prof.items() if v[1])
print(f"lib={os.path.basename(cw.lib_path())} alg={a.alg} comp={a.comp or '-'} bs={a.bs} nb={a.nb} data={a.data}: "
      f"{ms:.3f} ms/pass  {gb / (ms / 1e3):.1f} GB/s{extra}", flush=True)

"""Developer aid (not collected by pytest): compress corpus blocks on the device, compare with the oracle and print the
first differing LZ4 sequence.   python tests/debug_lz4_diff.py <block_bytes> [file ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import compute_war_amd as cw  # noqa: E402
import oracle as O  # noqa: E402


def seqs(c):
    ip, n, pos, out = 0, len(c), 0, []
    while ip < n:
        tok = c[ip]; ip += 1
        lit = tok >> 4
        if lit == 15:
            while True:
                b = c[ip]; ip += 1; lit += b
                if b != 255:
                    break
        ip += lit
        if ip >= n:
            out.append((pos, lit, 0, 0)); break
        off = c[ip] | (c[ip + 1] << 8); ip += 2
        ml = tok & 15
        if ml == 15:
            while True:
                b = c[ip]; ip += 1; ml += b
                if b != 255:
                    break
        out.append((pos, lit, off, ml + 4))
        pos += lit + ml + 4
    return out


bs = int(sys.argv[1])
files = sys.argv[2:] or ["alice29.txt", "kennedy.xls", "ptt5", "sum"]
cw.init(0)
pieces, names = [], []
for f in files:
    name, _, cut = f.partition(":")   # "name:bytes" takes a prefix; "zeros:bytes" is that many zero bytes
    d = bytes(int(cut)) if name == "zeros" else open(os.path.join(ROOT, "tests/golden/corpus/canterbury", name), "rb").read()
    if cut and name != "zeros":
        d = d[: int(cut)]
    d = d[: len(d) // bs * bs]
    pieces.append(d)
    names += [f"{name}[{i}]" for i in range(len(d) // bs)]
d = b"".join(pieces)                 # ONE call over everything: which blocks share a wavefront matters for some bugs
nb = len(d) // bs
sizes, payload = cw.compress_blocks("lz4", np.frombuffer(d, dtype=np.uint8), bs)
bad = 0
for i in range(nb):
    want = O.lz4_compress(d[i * bs:(i + 1) * bs])
    got = payload[i, : sizes[i]].tobytes()
    if got != want:
        bad += 1
        sg, sw = seqs(got), seqs(want)
        k = next((j for j in range(min(len(sg), len(sw))) if sg[j] != sw[j]), min(len(sg), len(sw)))
        print(f"block {i} = {names[i]}: size {len(got)} vs {len(want)}; seq {k} of {len(sw)}: got {sg[max(0,k-2):k+2]} want {sw[max(0,k-2):k+2]}", flush=True)
        if bad >= 8:
            break
print("mismatching blocks:", bad, "of", nb, "| kernels:", cw.profile_kernels()["codec"])

#!/usr/bin/env python3
"""Table probes, sequences and candidate fetches per block of the LZ4 fast parser / liblzf on a file (CPU, pure Python):
the per-block work counts that tools/random_line.hip's probe rates are compared with.  The emulation's
compressed sizes are checked against the oracle's in tests/test_tools.py."""
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def lz4_counts(b):
    n = len(b)
    mflimit, matchlimit = n - 12, n - 5
    tab = {}
    rd = lambda p: b[p:p + 4]
    h = lambda p: ((struct.unpack_from("<I", b, p)[0] * 2654435761) & 0xFFFFFFFF) >> 19
    probes = seqs = cand_equal_fp = 0
    out = 0
    anchor, ip = 0, 0
    tab[h(0)] = 0
    ip = 1
    done = False
    while not done:
        fwd, step, nb = ip, 1, 64
        while True:
            ip = fwd
            fwd += step
            step = nb >> 6
            nb += 1
            if fwd > mflimit:
                done = True
                break
            hv = h(ip)
            m = tab.get(hv, 0)
            probes += 1
            tab[hv] = ip
            if rd(m) == rd(ip):
                cand_equal_fp += 1
                break
        if done:
            break
        while True:
            while ip > anchor and m > 0 and b[ip - 1] == b[m - 1]:
                ip -= 1
                m -= 1
            lit = ip - anchor
            out += 1 + (0 if lit < 15 else 1 + (lit - 15) // 255) + lit + 2
            ml = 0
            while ip + 4 + ml < matchlimit and b[ip + 4 + ml] == b[m + 4 + ml]:
                ml += 1
            out += 0 if ml < 15 else 1 + (ml - 15) // 255
            seqs += 1
            ip += 4 + ml
            anchor = ip
            if ip > mflimit:
                done = True
                break
            tab[h(ip - 2)] = ip - 2
            hv = h(ip)
            m = tab.get(hv, 0)
            probes += 1
            tab[hv] = ip
            if rd(m) == rd(ip):
                cand_equal_fp += 1
                continue
            ip += 1
            break
    lit = n - anchor
    out += 1 + (0 if lit < 15 else 1 + (lit - 15) // 255) + lit
    return probes, seqs, out


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests/golden/corpus/canterbury/lcet10.txt")
    bs = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    data = open(path, "rb").read()
    nb = min(len(data) // bs, 4)
    tp = ts = to = 0
    for i in range(nb):
        blk = data[i * bs:(i + 1) * bs]
        p, s, o = lz4_counts(blk)
        tp, ts, to = tp + p, ts + s, to + o
    print(f"{os.path.basename(path)} {bs} B blocks x {nb}: LZ4 {tp / nb:.0f} table probes, {ts / nb:.0f} sequences per block, "
          f"{to / nb:.0f} B out")


if __name__ == "__main__":
    main()

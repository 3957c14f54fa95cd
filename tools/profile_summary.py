#!/usr/bin/env python3
"""Turns the raw output of tools/collect_profiles.sh (gpurun_out/<tag>/) into the tracked evidence under profiles/:
  profiles/<tag>_bench_default.json               the default bench line (headline + legs + host_path)
  profiles/<tag>_<leg>_kernel_stats.csv           rocprofv3 --kernel-trace --stats, cw:: kernels only, one file per command
  profiles/<tag>_<leg>_bench.json                 the bench line of that profiled command
  profiles/<tag>_pmc_summary.txt                  FETCH_SIZE (x2, gfx950 correction for 16 B/lane loads) / WRITE_SIZE per kernel and block
  profiles/traffic.json                           HBM bytes per step of the dominant kernel of every leg (bench.py reads it)
usage: tools/profile_summary.py <tag>
"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LEGS = {"headline": ("skein512", "lz4", 65536, 1 << 20, "random"),
        "mixed": ("skein512", "lz4", 65536, 65536, "mixed"),
        "corpus_skein512_lz4": ("skein512", "lz4", 65536, 65536, "corpus"),
        "corpus_skein512_lz4_16g": ("skein512", "lz4", 65536, 262144, "corpus"),
        "corpus_skein512_lz4_3233": ("skein512", "lz4", 65536, 3233, "corpus"),
        "corpus_skein512_lz4_51728": ("skein512", "lz4", 4096, 51728, "corpus"),
        "corpus_skein256_lz4_4k": ("skein", "lz4", 4096, 1 << 20, "corpus"),
        "corpus_sha256_lzf_4k": ("sha256mb", "lzf", 4096, 1 << 20, "corpus"),
        "corpus_sha256_lzf_64k": ("sha256mb", "lzf", 65536, 65536, "corpus"),
        "corpus_sha256_lzf_64k_16g": ("sha256mb", "lzf", 65536, 262144, "corpus")}


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0]


def pmc(path):
    """sum of the counter per kernel name over the run; (sum, dispatches)"""
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        if k.startswith("cw::"):
            agg[k][0] += float(r["Counter_Value"])
            agg[k][1] += 1
    return agg


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
    shutil.copy(os.path.join(src, "bench_default.json"), os.path.join(dst, f"{tag}_bench_default.json"))
    traffic = {"_note": "HBM bytes per STEP of a leg's kernels from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes; "
                        "tools/collect_profiles.sh: bench.py --no-legs --steps 1 --warmup 0 at the leg's own size), summed over the launches "
                        "of the step. FETCH_SIZE is doubled (gfx950 counts 128-B requests of 16 B/lane streaming loads as 64 B, "
                        "MI355X_MICROARCH.md); for scattered narrow loads (the parse kernels' candidates) that correction is an upper bound. "
                        "Counter collection serialises kernels: in the legs whose parsers run side by side the kernel that starts first "
                        "takes every block but what it is told to leave, so those sums describe that division of work, not the "
                        f"timed one. Raw per-kernel sums: profiles/{tag}_pmc_summary.txt"}
    lines = ["(rocprofv3 --pmc serialises kernels: where parsers run SIDE BY SIDE on one queue -- round 3: every LZ4 leg -- the kernel that starts",
             " first takes every block but what it is told to leave; per-block figures divide by all blocks of the step)", ""]
    for leg, (h, c, bs, nb, kind) in LEGS.items():
        ks = os.path.join(src, f"prof_{leg}", "p_kernel_stats.csv")
        if os.path.exists(ks):
            rows = [r for r in csv.reader(open(ks))]
            with open(os.path.join(dst, f"{tag}_{leg}_kernel_stats.csv"), "w", newline="") as f:
                w = csv.writer(f)
                w.writerow(rows[0])
                for r in rows[1:]:
                    if "cw::" in r[0]:
                        w.writerow(r)
            shutil.copy(os.path.join(src, f"prof_{leg}.json"), os.path.join(dst, f"{tag}_{leg}_bench.json"))
        f_p, w_p = (os.path.join(src, f"pmc_{leg}_{cn}", "p_counter_collection.csv") for cn in ("FETCH_SIZE", "WRITE_SIZE"))
        if not (os.path.exists(f_p) and os.path.exists(w_p)):
            continue
        fa, wa = pmc(f_p), pmc(w_p)
        lines.append(f"== {leg}: {h}+{c}, {nb} x {bs} B {kind} blocks, one step (bench.py --no-legs --steps 1 --warmup 0 ...)")
        lines.append(f"{'kernel':58s} {'launches':>8s} {'FETCH_SIZE raw B/block':>24s} {'x2':>12s} {'WRITE_SIZE B/block':>20s}")
        hash_b = comp_b = 0.0
        for k in sorted(set(fa) | set(wa)):
            fb, wb = fa[k][0] * 1024 / nb, wa[k][0] * 1024 / nb
            lines.append(f"{k:58s} {max(fa[k][1], wa[k][1]):8d} {fb:24,.0f} {2 * fb:12,.0f} {wb:20,.0f}")
            if "skein" in k or "sha256" in k:
                hash_b += (2 * fa[k][0] + wa[k][0]) * 1024
            elif "gen_" not in k and "sum_sizes" not in k and "decompress" not in k:   # the decoders belong to the post-run check, not to a step
                comp_b += (2 * fa[k][0] + wa[k][0]) * 1024
        lines.append("")
        traffic[f"hash:{h}:{bs}:{nb}:{kind}"] = int(hash_b)
        traffic[f"comp:{c}:{bs}:{nb}:{kind}"] = int(comp_b)
    open(os.path.join(dst, f"{tag}_pmc_summary.txt"), "w").write("\n".join(lines) + "\n")
    json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Summarise tools/pmc_parse.sh output: per kernel (name filter) average counter values per launch (and per block, given the
number of blocks of the launch), plus the ratios the SQ counters are read for."""
import collections
import csv
import glob
import sys

tag, filt = sys.argv[1], sys.argv[2]
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 0
agg, cnt = collections.defaultdict(float), collections.Counter()
for f in sorted(glob.glob(f"gpurun_out/pmc_{tag}_*/p_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if filt in r["Kernel_Name"]:
            k = (r["Kernel_Name"].replace("void ", "").split("(")[0][-44:], r["Counter_Name"])
            agg[k] += float(r["Counter_Value"])
            cnt[k] += 1
per = collections.defaultdict(dict)
for (kern, c), v in agg.items():
    per[kern][c] = v / cnt[(kern, c)]
for kern in sorted(per):
    d = per[kern]
    print(f"{kern}  ({int(max(cnt[(kern, c)] for c in d))} launches seen per pass)")
    for c in sorted(d):
        extra = f"   {d[c] / nb:12.1f} per block" if nb and c.startswith("SQ_INSTS") else ""
        print(f"    {c:26s} {d[c]:14.5g}{extra}")
    wc = d.get("SQ_WAVE_CYCLES")
    if wc:
        parts = [f"{n} {d[c] / wc:.3f}" for n, c in (("waiting (s_waitcnt etc.)", "SQ_WAIT_ANY"), ("issue-stalled", "SQ_WAIT_INST_ANY"),
                                                     ("VALU active", "SQ_ACTIVE_INST_VALU"), ("scalar active", "SQ_ACTIVE_INST_SCA"),
                                                     ("LDS active", "SQ_ACTIVE_INST_LDS")) if c in d]
        print("    share of wavefront cycles: " + ", ".join(parts))
    if "SQ_WAVES" in d and "SQ_WAVE_CYCLES" in d and "SQ_BUSY_CYCLES" in d and d["SQ_BUSY_CYCLES"]:
        print(f"    wavefront cycles / SQ busy cycles = {d['SQ_WAVE_CYCLES'] / d['SQ_BUSY_CYCLES']:.2f} (resident wavefronts per SQ while busy, as the counters scale them)")

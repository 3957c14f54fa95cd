#!/usr/bin/env python3
"""Summarise tools/pmc_parse.sh output: per kernel (name filter) average counter values per launch."""
import collections
import csv
import glob
import sys

tag, filt = sys.argv[1], sys.argv[2]
agg, cnt = collections.defaultdict(float), collections.Counter()
for f in sorted(glob.glob(f"gpurun_out/pmc_{tag}_*/p_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if filt in r["Kernel_Name"]:
            k = (r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])
            agg[k] += float(r["Counter_Value"])
            cnt[k] += 1
for k in sorted(agg):
    print(f"{k[0]:42s} {k[1]:26s} {agg[k] / cnt[k]:.4g}")

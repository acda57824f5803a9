#!/usr/bin/env python3
"""Summarise a tools/prof_celt.sh output directory: per-kernel average duration from the
kernel-trace stats and per-launch averages of every PMC counter collected.
usage: pmc_summary.py <dir>"""
import collections
import csv
import glob
import os
import re
import sys


def short(name):
    m = re.search(r"ca::(\w+)", name)
    return m.group(1) if m else None


def main(d):
    for f in glob.glob(os.path.join(d, "stats", "**", "*_kernel_stats.csv"), recursive=True):
        print("# kernel-trace stats (ns)")
        for r in csv.DictReader(open(f)):
            k = short(r["Name"])
            if k:
                print("%-28s calls=%s avg_ns=%s min_ns=%s max_ns=%s pct=%s" % (
                    k, r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["Percentage"]))
    print("# PMC, average per launch")
    for sub in ("pmc1", "pmc2", "pmc3", "pmc_fetch", "pmc_write"):
        agg = collections.defaultdict(list)
        for f in glob.glob(os.path.join(d, sub, "**", "*_counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if k:
                    agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(agg.items()):
            print("%-28s %-22s launches=%d avg=%.5g" % (k, c, len(v), sum(v) / len(v)))


if __name__ == "__main__":
    main(sys.argv[1])

#!/usr/bin/env python3
"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE csv output (two separate passes, as
MI355X_MICROARCH.md §HBM prescribes) into per-launch HBM traffic per kernel.

gfx950 corrections from the guide: counters are in KiB; FETCH_SIZE reports exactly half of the bytes
of a wide (16 B/lane) coalesced streaming read, so it is doubled; WRITE_SIZE is exact for 16 B/lane
streaming stores. Both kernels here stream with 16 B/lane, the calibrated pattern.

usage: pmc_traffic.py <fetch_dir> <write_dir> [out.json]
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def per_kernel(d, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def short(name):
    m = re.search(r"ca::(\w+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name


if __name__ == "__main__":
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        if "ca::" not in k:
            continue
        out[short(k)] = int((2.0 * fetch.get(k, 0.0) + write.get(k, 0.0)) * 1024)
    js = json.dumps(out, indent=1, sort_keys=True)
    print(js)
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(js + "\n")

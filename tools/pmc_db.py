#!/usr/bin/env python3
"""Turn the PMC passes of a tools/prof_celt.sh output directory into profiles/pmc.json: per kernel the per-launch
averages of the instruction / cycle counters bench.py's VALU-issue roofline needs (SQ_INSTS_VALU = vector wave-instructions
per launch, SQ_WAVES, SQ_WAVE_CYCLES, SQ_BUSY_CYCLES, SQ_WAIT_ANY, SQ_ACTIVE_INST_VALU ...) plus the batch size they were
collected at (units_per_launch), so the count can be scaled to the batch bench.py runs.

usage: pmc_db.py <prof dir> <units_per_launch> [out.json] [prefix]   (merges into an existing out.json; the passes are read
from <prof dir>/<prefix>pmc1 ... pmc4)"""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name):
    m = re.search(r"ca::(\w+)", name)
    return m.group(1) if m else None


def main(d, units, out=None, prefix=""):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for sub in ("pmc1", "pmc2", "pmc3", "pmc4"):
        for f in glob.glob(os.path.join(d, prefix + sub, "**", "*_counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if k:
                    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    db = {}
    if out and os.path.exists(out):
        db = json.load(open(out))
    for k, cs in agg.items():
        db[k] = {c: sum(v) / len(v) for c, v in cs.items()}
        db[k]["units_per_launch"] = units
    js = json.dumps(db, indent=1, sort_keys=True)
    if out:
        open(out, "w").write(js + "\n")
    else:
        print(js)


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), sys.argv[3] if len(sys.argv) > 3 else None, sys.argv[4] if len(sys.argv) > 4 else "")

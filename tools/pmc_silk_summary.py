#!/usr/bin/env python3
"""Per-kernel issue / wait picture of the SILK chain from the PMC passes tools/round_profile_silk.sh leaves in
gpurun_out/<tag>/<workload>_pmc1, _pmc2, _pmc4: per wavefront the VALU / SALU / LDS / vector-memory instruction counts and
the share of the wavefront's cycles in which it issued (VALU, any) or waited. SQ_WAVE_CYCLES, SQ_ACTIVE_INST_* and SQ_WAIT_*
all count quad-cycles, so their ratios are plain fractions. usage: pmc_silk_summary.py <dir> [workload ...] -> JSON on stdout"""
import collections
import csv
import glob
import json
import os
import re
import sys


def main(d, workloads):
    out = {}
    for w in workloads:
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for sub in ("pmc1", "pmc2", "pmc4"):
            for f in glob.glob(os.path.join(d, "%s_%s" % (w, sub), "**", "*_counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    m = re.search(r"(?:ca::|anonymous namespace\)::)(\w+)", r["Kernel_Name"])
                    if m:
                        agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in agg.items():
            c = collections.defaultdict(float, {n: sum(v) / len(v) for n, v in cs.items()})
            W = c["SQ_WAVES"]
            if not W or not c["SQ_WAVE_CYCLES"]:
                continue
            wc = c["SQ_WAVE_CYCLES"] / W
            out["%s/%s" % (w, k)] = {
                "waves": int(W), "wave_quad_cycles": round(wc),
                "valu_per_wave": round(c["SQ_INSTS_VALU"] / W), "salu_per_wave": round(c["SQ_INSTS_SALU"] / W),
                "lds_per_wave": round(c["SQ_INSTS_LDS"] / W), "vmem_rd_per_wave": round(c["SQ_INSTS_VMEM_RD"] / W),
                "vmem_wr_per_wave": round(c["SQ_INSTS_VMEM_WR"] / W),
                "valu_active_frac": round(c["SQ_ACTIVE_INST_VALU"] / W / wc, 3), "any_active_frac": round(c["SQ_ACTIVE_INST_ANY"] / W / wc, 3),
                "wait_any_frac": round(c["SQ_WAIT_ANY"] / W / wc, 3), "lds_active_frac": round(c["SQ_ACTIVE_INST_LDS"] / W / wc, 3),
                "lds_bank_conflict_frac": round(c["SQ_LDS_BANK_CONFLICT"] / W / wc, 3), "ifetch_per_wave": round(c["SQ_IFETCH"] / W),
            }
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:] or ["silk_frames", "silk_nlsf", "silk_lpc"])

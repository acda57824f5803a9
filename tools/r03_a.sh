#!/bin/bash
# GPU box, round 3 call A: suite, headline bench, MDCT at two sizes / two register budgets, LDS-conflict PMC pass.
TAG=${1:-r03_a}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log; tail -5 $O/pytest_gpu.log
timeout -k 10 300 python3 bench.py > $O/bench.json 2> $O/bench.err && cut -c1-400 $O/bench.json
for occ in 0 8; do
  for f in 4096 65536; do
    OPUSGPU_MDCT_OCC=$occ timeout -k 10 200 python3 bench.py --workload mdct --frames $f --no-cpu-baseline > $O/bench_mdct_occ${occ}_$f.json 2>> $O/bench.err && python3 - <<PY
import json
d = json.load(open("$O/bench_mdct_occ${occ}_$f.json"))
r = d["roofline"]
print("mdct occ $occ frames $f: %.1f M frames/s  %s %.5f ms other %.5f ms  frac %.4f" % (d["value"] / 1e6, r["kernel"], r["avg_launch_ms"], r["other_kernel_ms"], r["frac"]))
PY
  done
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_celt -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity > $O/bench_prof.json 2> $O/prof.err
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA --output-format csv -d $O/pmc2_celt -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity > /dev/null 2>> $O/prof.err
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --output-format csv -d $O/pmc2_mdct -- python3 $R/bench.py --workload mdct --steps 2 --warmup 0 --no-cpu-baseline --no-parity > /dev/null 2>> $O/prof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_mdct -- python3 $R/bench.py --workload mdct --steps 20 --warmup 2 --no-cpu-baseline --no-parity > /dev/null 2>> $O/prof.err
python3 - <<PY
import csv, glob, collections
for tag in ("pmc2_celt", "pmc2_mdct"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % tag, recursive=True):
        for row in csv.DictReader(open(f)):
            agg[row["Kernel_Name"].split("(")[0][:60]][row["Counter_Name"]] += float(row["Counter_Value"])
    for k, v in sorted(agg.items()):
        if v.get("SQ_ACTIVE_INST_LDS"):
            print("%s %-58s conflict/active_lds %.2f  wait_any/active %.2f" % (tag, k, v["SQ_LDS_BANK_CONFLICT"] / v["SQ_ACTIVE_INST_LDS"], v["SQ_WAIT_ANY"] / max(v["SQ_ACTIVE_INST_ANY"], 1)))
for f in glob.glob("$O/stats_*/**/*kernel_stats.csv", recursive=True):
    print(f.split("/")[-3] if "/" in f else f)
    for i, row in enumerate(csv.DictReader(open(f))):
        if i < 8: print("   %-70s calls %s avg %.1f us  %s%%" % (row["Name"][:70], row["Calls"], float(row["AverageNs"]) / 1e3, row["Percentage"]))
PY
cd $R && timeout -k 10 200 python3 tools/stage_profile.py 16384 noise lane > $O/stage_profile.txt 2>&1; tail -4 $O/stage_profile.txt | cut -c1-1500

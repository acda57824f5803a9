#!/bin/bash
# GPU box, round 3 call C: MDCT pre-rotation variants at both sizes, headline after the band-energy fix, front2 LDS conflict ratio.
TAG=${1:-r03_c}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_mdct_gpu.py tests/test_encode_gpu.py tests/test_decode_gpu.py tests/test_hooks_gpu.py -q -m gpu -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
for pre in 1 2 4; do
  for f in 4096 65536; do
    OPUSGPU_MDCT_PRE=$pre timeout -k 10 200 python3 bench.py --workload mdct --frames $f --no-cpu-baseline > $O/bench_mdct_pre${pre}_$f.json 2>> $O/bench.err && python3 - <<PY
import json
d = json.load(open("$O/bench_mdct_pre${pre}_$f.json"))
r = d["roofline"]
print("mdct pre $pre frames $f: %.1f M frames/s  %s %.5f ms other %.5f ms  frac %.4f" % (d["value"] / 1e6, r["kernel"], r["avg_launch_ms"], r["other_kernel_ms"], r["frac"]))
PY
  done
done
timeout -k 10 300 python3 bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err && python3 - <<PY
import json
d = json.load(open("$O/bench.json"))
print("celt: %.3f M frames/s, %.3f ms/step" % (d["value"] / 1e6, d["ms_per_step"]), [(k["kernel"], k["avg_launch_ms"]) for k in d["roofline"]["kernels"]])
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA --output-format csv -d $O/pmc2_celt -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity > /dev/null 2>> $O/prof.err
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("$O/pmc2_celt/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"].split("(")[0][:60]][row["Counter_Name"]] += float(row["Counter_Value"])
for k, v in sorted(agg.items()):
    if v.get("SQ_ACTIVE_INST_LDS"):
        print("%-58s conflict/active_lds %.2f  wait_any/active %.2f" % (k, v["SQ_LDS_BANK_CONFLICT"] / v["SQ_ACTIVE_INST_LDS"], v["SQ_WAIT_ANY"] / max(v["SQ_ACTIVE_INST_ANY"], 1)))
PY

#!/bin/bash
# GPU box, round 3 call E: after the cheaper swizzle addressing + band energies back to chunks.
TAG=${1:-r03_e}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_mdct_gpu.py tests/test_encode_gpu.py tests/test_decode_gpu.py tests/test_hooks_gpu.py -q -m gpu -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
for f in 4096 65536; do
  timeout -k 10 200 python3 bench.py --workload mdct --frames $f --no-cpu-baseline > $O/bench_mdct_$f.json 2>> $O/bench.err && python3 - <<PY
import json
d = json.load(open("$O/bench_mdct_$f.json"))
r = d["roofline"]
print("mdct frames $f: %.1f M frames/s  %s %.5f ms other %.5f ms  frac %.4f" % (d["value"] / 1e6, r["kernel"], r["avg_launch_ms"], r["other_kernel_ms"], r["frac"]))
PY
done
timeout -k 10 300 python3 bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err && python3 - <<PY
import json
d = json.load(open("$O/bench.json"))
print("celt: %.3f M frames/s, %.3f ms/step" % (d["value"] / 1e6, d["ms_per_step"]), [(k["kernel"], k["avg_launch_ms"]) for k in d["roofline"]["kernels"]])
PY
timeout -k 10 200 python3 bench.py --workload decode --no-cpu-baseline > $O/bench_decode.json 2>> $O/bench.err && cut -c1-220 $O/bench_decode.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_WAVES SQ_INSTS_LDS --output-format csv -d $O/pmc_mdct -- python3 $R/bench.py --workload mdct --steps 2 --warmup 0 --no-cpu-baseline --no-parity > /dev/null 2>> $O/prof.err
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("$O/pmc_mdct/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][:50]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        if row["Counter_Name"] == "SQ_INSTS_VALU": cnt[k] += 1
for k, v in agg.items():
    if "mdct" in k:
        n = cnt[k]
        print(k, "VALU insts per transform %.0f, LDS insts %.0f, conflict/active_lds %.2f" % (v["SQ_INSTS_VALU"] / n / 8192, v["SQ_INSTS_LDS"] / n / 8192, v["SQ_LDS_BANK_CONFLICT"] / max(v["SQ_ACTIVE_INST_LDS"], 1)))
PY

#!/bin/bash
# GPU box, round 3 call B: suite, headline with the private-memory-free back kernel + one-channel front2, full profile of the headline.
TAG=${1:-r03_b}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log; tail -5 $O/pytest_gpu.log
timeout -k 10 300 python3 bench.py > $O/bench.json 2> $O/bench.err && python3 - <<PY
import json
d = json.load(open("$O/bench.json"))
print("celt: %.3f M frames/s, %.3f ms/step" % (d["value"] / 1e6, d["ms_per_step"]), [(k["kernel"], k["avg_launch_ms"]) for k in d["roofline"]["kernels"]], "vs cpu", d.get("vs_cpu_baseline"))
PY
for w in 12 16 18; do
  OPUSGPU_FRONT2_WAVES=$w timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-parity > $O/bench_f2w$w.json 2>> $O/bench.err && python3 - <<PY
import json
d = json.load(open("$O/bench_f2w$w.json"))
print("front2 waves/CU $w: %.3f ms/step" % d["ms_per_step"], [(k["kernel"], k["avg_launch_ms"]) for k in d["roofline"]["kernels"]])
PY
done
timeout -k 10 200 python3 bench.py --workload celt_streams --no-cpu-baseline > $O/bench_streams.json 2>> $O/bench.err && cut -c1-200 $O/bench_streams.json
bash tools/prof_celt.sh $TAG/celt > $O/celt_summary_stdout.txt 2>&1; tail -30 $O/celt_summary_stdout.txt
python3 tools/pmc_traffic.py $O/celt/pmc_fetch $O/celt/pmc_write $O/traffic_celt.json
python3 tools/pmc_db.py $O/celt 65536 $O/pmc_celt.json > /dev/null
cd $R && timeout -k 10 200 python3 tools/stage_profile.py 16384 noise lane > $O/stage_profile.txt 2>&1; tail -3 $O/stage_profile.txt | cut -c1-1200
cd $R && timeout -k 10 400 python3 bench.py --workload silk_frames_cbr --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_silk_cbr.json 2> $O/bench_silk_cbr.err; python3 - <<PY
import json
try:
    d = json.load(open("$O/bench_silk_cbr.json"))
    print("silk_frames_cbr: %.3f M frames/s, %.2f ms/step, parity %s" % (d["value"] / 1e6, d["ms_per_step"], d["parity_checked"]))
except Exception as e:
    print("silk_frames_cbr failed", e); print(open("$O/bench_silk_cbr.err").read()[-1500:])
PY

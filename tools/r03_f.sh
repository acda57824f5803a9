#!/bin/bash
# GPU box, round 3 call F: MDCT start-up stagger sweep (config #2 size), del_dec half-filled wavefronts.
TAG=${1:-r03_f}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout -k 10 300 python3 -m pytest tests/test_mdct_gpu.py -q -m gpu -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log; tail -2 $O/pytest_gpu.log
for stg in 0 1 2 3 4 6 8; do
  OPUSGPU_MDCT_STAGGER=$stg timeout -k 10 200 python3 bench.py --workload mdct --frames 4096 --no-cpu-baseline --no-parity > $O/bench_mdct_stg$stg.json 2>> $O/bench.err && python3 - <<PY
import json
d = json.load(open("$O/bench_mdct_stg$stg.json"))
r = d["roofline"]
print("mdct 4096 stagger $stg: %.1f M frames/s  %s %.5f ms other %.5f ms  frac %.4f" % (d["value"] / 1e6, r["kernel"], r["avg_launch_ms"], r["other_kernel_ms"], r["frac"]))
PY
done
OPUSGPU_MDCT_STAGGER=3 timeout -k 10 200 python3 bench.py --workload mdct --frames 65536 --no-cpu-baseline --no-parity > $O/bench_mdct_big_stg3.json 2>> $O/bench.err && cut -c1-150 $O/bench_mdct_big_stg3.json
for l in 64 32; do
  OPUSGPU_DD_LANES=$l timeout -k 10 300 python3 bench.py --workload silk_deldec --no-cpu-baseline > $O/bench_dd_$l.json 2>> $O/bench.err && python3 - <<PY
import json
d = json.load(open("$O/bench_dd_$l.json"))
print("silk_deldec lanes $l: %.2f M records/s %.3f ms/step parity %s" % (d["value"] / 1e6, d["ms_per_step"], d["parity_checked"]))
PY
done

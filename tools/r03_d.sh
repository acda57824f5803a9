#!/bin/bash
# GPU box, round 3 call D: SILK streams mode (test + bench), headline after the DPP band-energy reductions, whole suite.
TAG=${1:-r03_d}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout -k 10 500 python3 -m pytest tests/test_silk_stream_gpu.py -q -m gpu -x > $O/pytest_stream.log 2>&1; echo "stream pytest rc=$?" | tee -a $O/pytest_stream.log; tail -15 $O/pytest_stream.log
timeout -k 10 300 python3 bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err && python3 - <<PY
import json
d = json.load(open("$O/bench.json"))
print("celt: %.3f M frames/s, %.3f ms/step" % (d["value"] / 1e6, d["ms_per_step"]), [(k["kernel"], k["avg_launch_ms"]) for k in d["roofline"]["kernels"]])
PY
timeout -k 10 600 python3 -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
timeout -k 10 600 python3 bench.py --workload silk_streams > $O/bench_silk_streams.json 2> $O/bench_silk_streams.err; python3 - <<PY
import json
try:
    d = json.load(open("$O/bench_silk_streams.json"))
    print("silk_streams: %.3f M frames/s, %.2f ms/step, parity %s, cpu %s" % (d["value"] / 1e6, d["ms_per_step"], d["parity_checked"], d.get("cpu_baseline", {}).get("value")))
except Exception as e:
    print("silk_streams failed", e); print(open("$O/bench_silk_streams.err").read()[-2500:])
PY

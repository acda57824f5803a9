#!/bin/bash
# quick check of the SILK chain kernels: parity tests, then per-kernel times of the chain and the stand-alone benches
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_g
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_silk_gpu.py tests/test_hooks_gpu.py tests/test_silk_stream_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
for w in silk_pred silk_frames; do
  timeout -k 10 600 python3 $R/bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err; echo "$w rc=$?"; cut -c1-330 $O/bench_$w.json
done
rm -rf $O/st
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python3 $R/bench.py --workload silk_frames --steps 3 --warmup 1 --no-cpu-baseline > $O/b_prof.json 2> $O/b_prof.err
python3 - <<PY
import csv,glob
for f in glob.glob("$O/st/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if float(r["AverageNs"]) > 20000: print("  ", r["Name"][:46], r["Calls"], round(float(r["AverageNs"])/1e6,3))
PY

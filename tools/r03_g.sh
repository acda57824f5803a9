#!/bin/bash
# quick check of the SILK prediction kernels: parity tests, then the three benches
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_g
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_silk_gpu.py tests/test_hooks_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for w in silk_nlsf silk_pred silk_frames; do
  timeout -k 10 600 python3 bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err; echo "$w rc=$?"; cut -c1-330 $O/bench_$w.json
done

#!/bin/bash
# decode workload at HEAD: bench line + kernel-trace stats (profiles/r03_s8)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_s8
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_decode_gpu.py tests/test_sharding_gpu.py -x -q -m gpu > $O/pytest_decode.log 2>&1; echo "pytest rc=$?"; tail -1 $O/pytest_decode.log
timeout -k 10 300 python3 bench.py --workload decode > $O/bench_decode.json 2> $O/bench_decode.err && cut -c1-200 $O/bench_decode.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_decode -- python3 $R/bench.py --workload decode --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_prof_decode.json 2> $O/prof_decode.err; echo "rc=$?"

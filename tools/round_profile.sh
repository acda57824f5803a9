#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): the round's evidence in one call -- GPU test suite, the default bench with the CPU
# baseline, kernel-trace stats + PMC passes of the default workload (tools/prof_celt.sh), kernel-trace stats of the
# decode and silk_deldec workloads. Output: gpurun_out/<tag>/.  usage: tools/round_profile.sh <tag>
TAG=${1:?tag}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout -k 10 300 python3 bench.py > $O/bench.json 2> $O/bench.err && cut -c1-160 $O/bench.json
bash tools/prof_celt.sh $TAG/celt > $O/celt_summary_stdout.txt 2>&1; tail -3 $O/celt_summary_stdout.txt
cd /tmp && export TMPDIR=/tmp
for w in decode silk silk_deldec mixed; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$w -- python3 $R/bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_prof_$w.json 2> $O/prof_$w.err
  echo "$w rc=$?"
done

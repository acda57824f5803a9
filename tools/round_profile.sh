#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): the round's evidence in one call -- GPU test suite, the default bench with the CPU
# baseline, kernel-trace stats + PMC passes of the default workload (tools/prof_celt.sh), PMC traffic of the MDCT workload,
# kernel-trace stats of every other workload. Output: gpurun_out/<tag>/.  usage: tools/round_profile.sh <tag>
TAG=${1:?tag}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout -k 10 300 python3 bench.py > $O/bench.json 2> $O/bench.err && cut -c1-160 $O/bench.json
# config #2 at its own size and at 65 536 frames (one round vs steady state: DESIGN section 4), CPU baseline included
timeout -k 10 200 python3 bench.py --workload mdct > $O/bench_mdct_4096.json 2>> $O/bench.err && cut -c1-160 $O/bench_mdct_4096.json
timeout -k 10 200 python3 bench.py --workload mdct --frames 65536 --no-cpu-baseline > $O/bench_mdct_65536.json 2>> $O/bench.err && cut -c1-160 $O/bench_mdct_65536.json
bash tools/prof_celt.sh $TAG/celt > $O/celt_summary_stdout.txt 2>&1; tail -3 $O/celt_summary_stdout.txt
python3 tools/pmc_traffic.py $O/celt/pmc_fetch $O/celt/pmc_write $O/traffic_celt.json > /dev/null
python3 tools/pmc_db.py $O/celt 65536 $O/pmc_celt.json
cd /tmp && export TMPDIR=/tmp
for w in mdct decode; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${w}_pmc_fetch -- python3 $R/bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline --no-parity > /dev/null 2>> $O/prof_$w.err &&
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${w}_pmc_write -- python3 $R/bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline --no-parity > /dev/null 2>> $O/prof_$w.err &&
  python3 $R/tools/pmc_traffic.py $O/${w}_pmc_fetch $O/${w}_pmc_write $O/traffic_$w.json > /dev/null
done
for w in mdct decode silk silk_deldec silk_lpc mixed celt_streams; do
  case $w in silk*|mixed) python3 $R/bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $O/prepare_$w.err || { echo "$w prepare failed"; continue; };; esac
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$w -- python3 $R/bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_prof_$w.json 2> $O/prof_$w.err
  echo "$w rc=$?"
done

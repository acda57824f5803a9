#!/bin/bash
# Runs ON THE GPU BOX (via gpurun), second half of a round's evidence: kernel-trace stats of the SILK analysis-chain workloads
# (each builds its corpus from the capture build of the reference first, outside any clock). usage: tools/round_profile_silk.sh <tag>
TAG=${1:?tag}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for w in silk_frames silk_frames_cbr silk_streams silk_analysis silk_pred silk_nlsf; do
  # corpus capture (worker interpreters) and any stale checker library are built by a PLAIN python3 run first: nothing
  # under the profiler may start child processes (bench.py refuses to, tests/silk_corpus.py under_profiler)
  python3 $R/bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_prepare_$w.json 2> $O/prepare_$w.err || { echo "$w prepare failed"; continue; }
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$w -- python3 $R/bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_prof_$w.json 2> $O/prof_$w.err
  echo "$w rc=$?"; cut -c1-200 $O/bench_prof_$w.json
done
w=silk_frames
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${w}_pmc_fetch -- python3 $R/bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline --no-parity > /dev/null 2>> $O/prof_pmc_$w.err &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${w}_pmc_write -- python3 $R/bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline --no-parity > /dev/null 2>> $O/prof_pmc_$w.err &&
python3 $R/tools/pmc_traffic.py $O/${w}_pmc_fetch $O/${w}_pmc_write $O/traffic_$w.json > /dev/null
# issue / wait / instruction-mix counters of the chain's kernels and of its two heaviest parts on their own (passes of at most
# eight counters, each its own run; tools/pmc_db.py <dir> <units> <out> <prefix> reads them)
pmc_passes() {
  local w=$1
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/${w}_pmc1 -- python3 $R/bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline --no-parity > /dev/null 2>> $O/prof_pmc_$w.err &&
  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA --output-format csv -d $O/${w}_pmc2 -- python3 $R/bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline --no-parity > /dev/null 2>> $O/prof_pmc_$w.err
  echo "pmc $w rc=$?"
}
python3 $R/bench.py --workload silk_lpc --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_prepare_silk_lpc.json 2> $O/prepare_silk_lpc.err   # its corpus, outside the profiler
pmc_passes silk_frames && pmc_passes silk_nlsf && pmc_passes silk_lpc
# instruction-fetch stalls (the analysis kernels are long straight-line code); last, since the counter names may not exist
w=silk_frames
rocprofv3 --pmc SQ_IFETCH SQ_WAIT_IFETCH SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d $O/${w}_pmc4 -- python3 $R/bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline --no-parity > /dev/null 2>> $O/prof_pmc_$w.err
echo "pmc4 rc=$?"

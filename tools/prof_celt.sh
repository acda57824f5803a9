#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): kernel-trace stats, then PMC passes (each its own run, never
# combined with a trace domain), for bench.py's default workload. Output: gpurun_out/<tag>/.
# usage: tools/prof_celt.sh <tag> [extra bench args]
TAG=${1:?tag}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity "$@" > $O/bench_prof.json 2> $O/prof.err &&
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --output-format csv -d $O/pmc1 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity "$@" > /dev/null 2>> $O/prof.err &&
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA --output-format csv -d $O/pmc2 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity "$@" > /dev/null 2>> $O/prof.err &&
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d $O/pmc3 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity "$@" > /dev/null 2>> $O/prof.err &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity "$@" > /dev/null 2>> $O/prof.err &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity "$@" > /dev/null 2>> $O/prof.err &&
python3 $R/tools/pmc_summary.py $O > $O/summary.txt && cat $O/summary.txt

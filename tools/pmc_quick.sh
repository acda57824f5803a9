#!/bin/bash
# Runs ON THE GPU BOX: one PMC pass (instruction counts) of the default bench. usage: tools/pmc_quick.sh <tag>
TAG=${1:?tag}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --output-format csv -d $O/pmc1 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $O/prof.err
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA --output-format csv -d $O/pmc2 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>> $O/prof.err
python3 $R/tools/pmc_summary.py $O > $O/summary.txt; grep -E "front1|front2" $O/summary.txt

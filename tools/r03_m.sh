#!/bin/bash
# decoder: PMC picture of its three kernels + the stage stamps of the lane kernel
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_m
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
w=decode
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/${w}_pmc1 -- python3 $R/bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline --no-parity > /dev/null 2>> $O/prof.err &&
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA --output-format csv -d $O/${w}_pmc2 -- python3 $R/bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline --no-parity > /dev/null 2>> $O/prof.err
echo "rc=$?"
python3 $R/tools/pmc_silk_summary.py $O decode
timeout -k 10 300 python3 $R/tools/stage_profile_decode.py > $O/stage_decode.txt 2>&1; tail -12 $O/stage_decode.txt

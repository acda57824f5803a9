#!/bin/bash
# PMC picture of the config-#4 kernels (silk_burg_modified + silk_NSQ)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_i
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
w=silk
python3 $R/bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline > $O/prep.json 2> $O/prep.err || exit 1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/${w}_pmc1 -- python3 $R/bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline --no-parity > /dev/null 2>> $O/prof.err &&
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA --output-format csv -d $O/${w}_pmc2 -- python3 $R/bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline --no-parity > /dev/null 2>> $O/prof.err &&
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TA_BUSY_avr TA_TA_BUSY_sum --output-format csv -d $O/${w}_pmc5 -- python3 $R/bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline --no-parity > /dev/null 2>> $O/prof.err
echo "rc=$?"
python3 $R/tools/pmc_silk_summary.py $O silk

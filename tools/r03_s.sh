#!/bin/bash
# back lane kernel at 64 frames per wavefront: stage stamps + a PMC pass of the wait / fetch / memory-instruction counters
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_s
mkdir -p $O
cd $R
timeout -k 10 300 python3 tools/stage_profile.py 65536 noise lane > $O/stage.txt 2>&1; tail -2 $O/stage.txt | head -1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_BRANCH SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_CYCLES_SALU --output-format csv -d $O/pmc_a -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity > /dev/null 2>> $O/prof.err || echo "pmc a failed"
rocprofv3 --pmc SQ_WAVES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_b -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity > /dev/null 2>> $O/prof.err || echo "pmc b failed"
python3 - <<PY
import csv, glob, collections
for tag in ("pmc_a", "pmc_b"):
    acc = collections.defaultdict(float)
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            if "back_lane" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"])
    print(tag, dict(acc))
PY

#!/bin/bash
# back lane kernel: where a wavefront waits -- in-flight levels of vector memory and LDS instructions (average latency =
# level / instructions), at 64 and 32 frames per wavefront
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_p
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $O/sq_counters.txt
for a in 64 32; do
  export OPUSGPU_LANE_FRAMES=$a
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_WAIT_ANY --output-format csv -d $O/pmc_$a -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity > /dev/null 2>> $O/prof.err || echo "pmc failed $a"
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU --output-format csv -d $O/pmcb_$a -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity > /dev/null 2>> $O/prof.err || echo "pmcb failed $a"
done
python3 - <<PY
import csv, glob, collections
for a in (64, 32):
    for tag in ("pmc", "pmcb"):
        acc = collections.defaultdict(float)
        for f in glob.glob("$O/%s_%d/**/*counter_collection.csv" % (tag, a), recursive=True):
            for r in csv.DictReader(open(f)):
                if "back_lane" in r["Kernel_Name"]:
                    acc[r["Counter_Name"]] += float(r["Counter_Value"])
        print(a, tag, dict(acc))
PY

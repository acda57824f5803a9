#!/bin/bash
# the mixed workload (114 688 CELT frames + 16 384 SILK records per rank) and streams mode at 64 / 32 frames per wavefront
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_v
mkdir -p $O
cd $R
python3 bench.py --workload mixed --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $O/prep.err
for a in 64 32; do
  export OPUSGPU_LANE_FRAMES=$a
  for w in mixed celt_streams; do
    timeout -k 10 300 python3 bench.py --workload $w --steps 8 --warmup 2 --no-cpu-baseline > $O/${w}_$a.json 2> $O/${w}_$a.err || exit 1
    python3 -c "
import json; d=json.loads(open('$O/${w}_$a.json').read().strip().splitlines()[-1]); print($a, '$w', d['value'], d['ms_per_step'], [(k['kernel'],k['avg_launch_ms']) for k in d['roofline'].get('kernels',[])][:3])"
  done
done

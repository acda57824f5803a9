#!/bin/bash
# frames per wavefront of the back lane kernel: 32 (two wavefronts per SIMD), 21 (three, 168 VGPRs), 16 (four, 128 VGPRs)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_k
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in "32 0" "21 0" "16 1" "32 0"; do
  set -- $cfg
  export OPUSGPU_LANE_FRAMES=$1
  if [ "$2" = "1" ]; then export OPUSGPU_LANE16_OCC4=1; else unset OPUSGPU_LANE16_OCC4; fi
  timeout -k 10 300 python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/b_$1.json 2> $O/b_$1.err || { echo "lane $1 failed"; tail -3 $O/b_$1.err; exit 1; }
  python3 - <<PY
import json
d=json.loads(open("$O/b_$1.json").read().strip().splitlines()[-1])
print("lanes $1:", d["value"], d["ms_per_step"], "parity", d.get("parity_checked"), [ (k["kernel"][:14], k["avg_launch_ms"]) for k in d["roofline"]["kernels"][:1]])
PY
done

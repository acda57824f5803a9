#!/bin/bash
# frames per wavefront of the back lane kernel at HEAD: 64 vs 32 (encoder and decoder)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_o
mkdir -p $O
cd $R
for a in ${LANES_LIST:-32 64}; do
  export OPUSGPU_LANE_FRAMES=$a
  timeout -k 10 200 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/enc_$a.json 2> $O/enc_$a.err || exit 1
  timeout -k 10 200 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload decode > $O/dec_$a.json 2> $O/dec_$a.err || exit 1
  python3 - <<PY
import json
for w in ("enc","dec"):
    d=json.loads(open("$O/%s_$a.json"%w).read().strip().splitlines()[-1]); print("$a",w,d["value"],d["ms_per_step"],d.get("parity_checked"),[(k["kernel"],k["avg_launch_ms"]) for k in d["roofline"].get("kernels", d["roofline"].get("other_kernels", []))], d["roofline"].get("avg_launch_ms"))
PY
done

#!/bin/bash
# encoder after a back-lane change: parity tests, bench at 64 and 32 frames per wavefront
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_r
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_encode_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
for a in ${LANES_LIST:-64 32}; do
  export OPUSGPU_LANE_FRAMES=$a
  timeout -k 10 200 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/enc_$a.json 2> $O/enc_$a.err || exit 1
  python3 - <<PY
import json
d=json.loads(open("$O/enc_$a.json").read().strip().splitlines()[-1]); print("$a",d["value"],d["ms_per_step"],d.get("parity_checked"),[(k["kernel"],k["avg_launch_ms"]) for k in d["roofline"].get("kernels", d["roofline"].get("other_kernels", []))], d["roofline"].get("avg_launch_ms"))
PY
done

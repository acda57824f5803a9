#!/bin/bash
# decoder after a lane-kernel change: parity tests, the decode bench, the stage stamps
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_n
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_decode_gpu.py tests/test_hooks_gpu.py -x -q -m gpu -k "decode or quant_all_bands or dec" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 python3 $R/bench.py --workload decode --steps 10 --warmup 2 --no-cpu-baseline > $O/b.json 2> $O/b.err || exit 1
python3 - <<PY
import json
d=json.loads(open("$O/b.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], "parity", d.get("parity_checked"), [(k["kernel"], k["avg_launch_ms"]) for k in d["roofline"].get("kernels", [])])
PY
timeout -k 10 300 python3 $R/tools/stage_profile_decode.py > $O/stage_decode.txt 2>&1; tail -1 $O/stage_decode.txt

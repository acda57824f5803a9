#!/bin/bash
# CELT encoder after a back-phase change: the encode parity tests, then the headline bench
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_l
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_encode_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
for i in 1 2; do
timeout -k 10 300 python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/b$i.json 2> $O/b$i.err || exit 1
python3 - <<PY
import json
d=json.loads(open("$O/b$i.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], "parity", d.get("parity_checked"), [(k["kernel"][5:16], k["avg_launch_ms"]) for k in d["roofline"]["kernels"]])
PY
done

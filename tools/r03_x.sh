#!/bin/bash
# HEAD check: the whole GPU suite, the default bench line, smoke()
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_x
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.log
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout -k 10 300 python3 bench.py > $O/bench.json 2> $O/bench.err && cut -c1-220 $O/bench.json

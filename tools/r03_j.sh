#!/bin/bash
# config #4 + del_dec after a kernel change: parity tests, bench lines, kernel times
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_j
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_silk_gpu.py tests/test_hooks_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
for w in silk silk_deldec; do
  python3 $R/bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $O/prep_$w.err
  rm -rf $O/st_$w
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$w -- python3 $R/bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline > $O/b_$w.json 2> $O/b_$w.err
  cut -c1-250 $O/b_$w.json
  python3 - <<PY
import csv,glob
for f in glob.glob("$O/st_$w/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "ca::" in r["Name"]: print("  ", r["Name"][:46], r["Calls"], round(float(r["AverageNs"])/1e6,3))
PY
done

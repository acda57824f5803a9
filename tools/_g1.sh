R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_full; mkdir -p $O
cd $R
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; tail -5 $O/tests.log

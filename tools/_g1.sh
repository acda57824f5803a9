R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_z3; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_hooks_gpu.py -x -q -m gpu -k "silk" > $O/tests.log 2>&1; tail -25 $O/tests.log

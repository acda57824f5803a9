R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_zb; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_hooks_gpu.py -x -q -m gpu -k "four_frame" > $O/tests.log 2>&1; tail -30 $O/tests.log

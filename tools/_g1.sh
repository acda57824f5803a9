R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_z9; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_silk_chain_gpu.py -x -q -m gpu > $O/tests.log 2>&1; tail -30 $O/tests.log

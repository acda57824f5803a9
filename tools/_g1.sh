R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_z5; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_silk_gpu.py -x -q -m gpu -k "noise_shape" > $O/tests.log 2>&1; tail -8 $O/tests.log

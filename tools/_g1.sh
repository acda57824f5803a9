R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_z; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_silk_gpu.py tests/test_hooks_gpu.py -x -q -m gpu > $O/tests.log 2>&1; tail -15 $O/tests.log
timeout -k 10 600 python3 bench.py --workload silk_nlsf > $O/nlsf.json 2> $O/nlsf.err; tail -c 1500 $O/nlsf.json; tail -3 $O/nlsf.err

cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_u; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_hooks_gpu.py -q -m gpu -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest_gpu.log

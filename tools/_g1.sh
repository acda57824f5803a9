cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02_c
timeout -k 10 900 python3 -m pytest tests -q -m gpu -x > gpurun_out/r02_c/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r02_c/pytest_gpu.log

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02_a
timeout -k 10 900 python3 -m pytest tests -q -m gpu -x > gpurun_out/r02_a/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r02_a/pytest_gpu.log
timeout -k 10 300 python3 bench.py > gpurun_out/r02_a/bench.json 2> gpurun_out/r02_a/bench.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/r02_a/bench.json
timeout -k 10 200 python3 tools/stage_profile.py 16384 noise lane > gpurun_out/r02_a/stage.txt 2>&1; echo "stage rc=$?"; tail -30 gpurun_out/r02_a/stage.txt

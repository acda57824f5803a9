cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_v; mkdir -p $O
CONCENTUS_BENCH_BACKEND=gloo CONCENTUS_BENCH_ONE_DEVICE=1 timeout -k 10 280 python3 bench.py --gpus 2 --frames 16384 --steps 3 --no-cpu-baseline > $O/bench_2r.json 2>$O/bench_2r.err; echo "rc=$?"; tail -3 $O/bench_2r.err | cut -c1-300; cut -c1-500 $O/bench_2r.json
CONCENTUS_BENCH_BACKEND=gloo CONCENTUS_BENCH_ONE_DEVICE=1 timeout -k 10 280 python3 bench.py --gpus 2 --workload mixed --frames 32768 --steps 2 --no-cpu-baseline > $O/bench_2r_mixed.json 2>$O/bench_2r_mixed.err; echo "rc=$?"; tail -3 $O/bench_2r_mixed.err | cut -c1-300; cut -c1-400 $O/bench_2r_mixed.json

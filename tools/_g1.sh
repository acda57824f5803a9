cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_r; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_decode_gpu.py tests/test_fuzz_gpu.py tests/test_c_host_gpu.py -q -m gpu -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
timeout -k 10 120 python3 bench.py --workload decode --no-cpu-baseline > $O/bench_dec.json 2>$O/bench_dec.err; python3 -c "
import json;d=json.load(open('$O/bench_dec.json'));print(d['value'], d['ms_per_step'], d['parity_checked'], [(k['kernel'][5:],k['avg_launch_ms']) for k in d['roofline']['kernels']])"

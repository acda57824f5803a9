cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_q; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_mdct_gpu.py tests/test_hooks_gpu.py tests/test_encode_gpu.py tests/test_decode_gpu.py -q -m gpu -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
timeout -k 10 120 python3 bench.py --workload mdct --no-cpu-baseline > $O/bench_mdct.json 2>$O/bench_mdct.err; python3 -c "
import json;d=json.load(open('$O/bench_mdct.json'));print(d['value'], d['ms_per_step'], d['parity_checked'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'], d['roofline']['other_kernel_ms'], d['roofline']['frac'])"
timeout -k 10 120 python3 bench.py --no-cpu-baseline --steps 5 > $O/bench.json 2>$O/bench.err; python3 -c "
import json;d=json.load(open('$O/bench.json'));print(d['value'], d['ms_per_step'], d['parity_checked'], [(k['kernel'][5:],k['avg_launch_ms']) for k in d['roofline']['kernels']])"

cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_k; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
OPUSGPU_LANE16_OCC4=1 OPUSGPU_LANE_FRAMES=16 timeout -k 10 120 python3 bench.py --no-cpu-baseline --steps 5 > $O/bench_16o4.json 2>$O/bench_16o4.err; python3 -c "
import json;d=json.load(open('$O/bench_16o4.json'));print('16o4', d['value'], d['ms_per_step'], d['parity_checked'], [(k['kernel'][5:],k['avg_launch_ms']) for k in d['roofline']['kernels']])"
bash tools/prof_celt.sh r02_k/celt > $O/celt_summary_stdout.txt 2>&1; grep back_lane $O/celt/summary.txt

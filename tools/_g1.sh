cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_h; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_encode_gpu.py tests/test_fuzz_gpu.py -q -m gpu -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
for a in 32 64; do OPUSGPU_LANE_FRAMES=$a timeout -k 10 120 python3 bench.py --no-cpu-baseline --steps 5 > $O/bench_$a.json 2>$O/bench_$a.err; python3 -c "
import json;d=json.load(open('$O/bench_$a.json'));print($a, d['value'], d['ms_per_step'], d['parity_checked'], [(k['kernel'][5:],k['avg_launch_ms']) for k in d['roofline']['kernels']])"; done
timeout -k 10 200 python3 tools/stage_profile.py 16384 noise lane > $O/stage.txt 2>&1; echo "stage rc=$?"; grep -A 25 "lane-per-frame back" $O/stage.txt | grep -E "pvq|tf_|total|pitch" | cut -c1-400

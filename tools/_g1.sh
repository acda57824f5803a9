R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_w; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for a in 32 64; do
export OPUSGPU_LANE_FRAMES=$a
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f$a -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity > /dev/null 2>> $O/prof.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w$a -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity > /dev/null 2>> $O/prof.err
python3 $R/tools/pmc_traffic.py $O/f$a $O/w$a $O/traffic_$a.json | grep back_lane
python3 $R/bench.py --steps 10 --no-cpu-baseline --no-parity 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print($a, d['value'], d['ms_per_step'], [(k['kernel'][5:],k['avg_launch_ms']) for k in d['roofline']['kernels']][:2])"
done

R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_za; mkdir -p $O
cd $R
timeout -k 10 1000 python3 bench.py --workload silk_frames > $O/frames.json 2> $O/frames.err; python3 -c "
import json;d=json.load(open('$O/frames.json'));print(d['value'], d['ms_per_step'], d['parity_checked'], d['cpu_baseline']['value'], d['vs_cpu_baseline'])"; tail -3 $O/frames.err

R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_z8; mkdir -p $O
cd $R
timeout -k 10 1000 python3 bench.py --workload silk_analysis > $O/ana.json 2> $O/ana.err; python3 -c "
import json;d=json.load(open('$O/ana.json'));print(d['value'], d['ms_per_step'], d['parity_checked'], d['cpu_baseline']['value'], d['vs_cpu_baseline']); print(d['roofline']['kernels'])"; tail -3 $O/ana.err

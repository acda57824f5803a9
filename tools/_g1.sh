R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_z2; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_silk_gpu.py -x -q -m gpu -k "pred_coefs or nlsfs" > $O/tests.log 2>&1; tail -15 $O/tests.log
timeout -k 10 600 python3 bench.py --workload silk_pred > $O/pred.json 2> $O/pred.err; python3 -c "
import json;d=json.load(open('$O/pred.json'));print(d['value'], d['ms_per_step'], d['parity_checked'], d['cpu_baseline']['value'], d['vs_cpu_baseline'])"; tail -3 $O/pred.err

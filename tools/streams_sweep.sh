set -e
mkdir -p gpurun_out/st
for n in 1 2 3; do
  timeout -k 10 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --streams $n > gpurun_out/st/enc_$n.json 2> gpurun_out/st/enc_$n.err
  python -c "
import json; d=json.load(open('gpurun_out/st/enc_$n.json')); print($n, d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'], [(k['kernel'],k['avg_launch_ms']) for k in d['roofline']['other_kernels']])"
done

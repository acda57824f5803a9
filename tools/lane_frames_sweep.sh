set -e
mkdir -p gpurun_out/lf
for a in 64 32 16; do
  export OPUSGPU_LANE_FRAMES=$a
  timeout -k 10 200 python -m pytest tests/test_encode_gpu.py tests/test_decode_gpu.py -x -q -m gpu > gpurun_out/lf/tests_$a.log 2>&1
  tail -1 gpurun_out/lf/tests_$a.log
  timeout -k 10 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/lf/enc_$a.json 2> gpurun_out/lf/enc_$a.err
  timeout -k 10 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --workload decode > gpurun_out/lf/dec_$a.json 2> gpurun_out/lf/dec_$a.err
  python - <<PY
import json
for w in ("enc","dec"):
    d=json.load(open("gpurun_out/lf/%s_$a.json"%w)); print("$a",w,d["value"],d["ms_per_step"],d.get("kernels"))
PY
done

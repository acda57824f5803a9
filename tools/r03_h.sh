#!/bin/bash
# experiment: partially filled wavefronts in the pitch / bits kernels; half batches of the CELT pipeline on two streams
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_h
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --workload silk_frames --steps 1 --warmup 0 --no-cpu-baseline > $O/prep.json 2> $O/prep.err || exit 1
for cfg in "64 64" "32 64" "32 32" "32 16" "16 32"; do
  set -- $cfg
  export OPUSGPU_SILK_LANES=$1 OPUSGPU_SILK_BITS_LANES=$2
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$1_$2 -- python3 $R/bench.py --workload silk_frames --steps 3 --warmup 1 --no-cpu-baseline > $O/b_$1_$2.json 2> $O/b_$1_$2.err || exit 1
  echo "== lanes pitch=$1 bits=$2"; python3 - <<PY
import csv,glob,json
d=json.loads(open("$O/b_$1_$2.json").read().strip().splitlines()[-1]); print(" ms_per_step", d["ms_per_step"], "parity", str(d.get("parity_checked"))[:60])
for f in glob.glob("$O/st_$1_$2/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pitch_lags" in r["Name"] or "encode_bits" in r["Name"]: print("  ", r["Name"][:40], round(float(r["AverageNs"])/1e6,3))
PY
done
unset OPUSGPU_SILK_LANES OPUSGPU_SILK_BITS_LANES
for cfg in "65536 1" "32768 1" "32768 2" "16384 4" "65536 2"; do
  set -- $cfg
  python3 $R/bench.py --frames $1 --streams $2 --steps 12 --warmup 3 --no-cpu-baseline --no-parity > $O/celt_$1_$2.json 2> $O/celt_$1_$2.err || exit 1
  echo "== celt frames=$1 streams=$2"; cut -c1-200 $O/celt_$1_$2.json
done

#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_t
mkdir -p $O
cd $R
for n in 1 2 3; do
  timeout -k 10 120 python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --streams $n > $O/enc_$n.json 2> $O/enc_$n.err || exit 1
  python3 -c "
import json; d=json.loads(open('$O/enc_$n.json').read().strip().splitlines()[-1]); print($n, d['value'], d['ms_per_step'], d.get('parity_checked'))"
done

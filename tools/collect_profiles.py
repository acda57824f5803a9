#!/usr/bin/env python3
"""Copy the judged summaries of a tools/round_profile.sh run from gpurun_out/<tag> (scratch) into profiles/<tag> (tracked):
the kernel-trace stats of every workload, the PMC summary of the default workload, the bench lines, the GPU test log; merge
the per-kernel HBM traffic and instruction counts into profiles/traffic.json and profiles/pmc.json (read by bench.py).
usage: collect_profiles.py <tag>"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles", tag)
os.makedirs(dst, exist_ok=True)


def cp(a, b):
    if os.path.exists(a):
        shutil.copy(a, os.path.join(dst, b))


cp(os.path.join(src, "pytest_gpu.log"), "pytest_gpu.log")
cp(os.path.join(src, "bench.json"), "bench.json")
cp(os.path.join(src, "bench_mdct_4096.json"), "bench_mdct_4096.json")
cp(os.path.join(src, "bench_mdct_65536.json"), "bench_mdct_65536.json")
cp(os.path.join(src, "celt", "summary.txt"), "summary.txt")
cp(os.path.join(src, "celt", "bench_prof.json"), "bench_under_rocprof.json")
for f in glob.glob(os.path.join(src, "celt", "stats", "**", "*_kernel_stats.csv"), recursive=True):
    cp(f, "kernel_stats.csv")
for w in ("mdct", "decode", "silk", "silk_deldec", "silk_lpc", "mixed", "celt_streams", "silk_frames", "silk_frames_cbr", "silk_streams", "silk_analysis", "silk_pred", "silk_nlsf"):
    cp(os.path.join(src, "bench_prof_%s.json" % w), "bench_under_rocprof_%s.json" % w)
    for f in glob.glob(os.path.join(src, "stats_%s" % w, "**", "*_kernel_stats.csv"), recursive=True):
        cp(f, "kernel_stats_%s.csv" % w)
if glob.glob(os.path.join(src, "silk_frames_pmc1")):
    import subprocess
    js = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "pmc_silk_summary.py"), src])
    open(os.path.join(dst, "pmc_silk.json"), "wb").write(js)
traffic = {}
tp = os.path.join(ROOT, "profiles", "traffic.json")
if os.path.exists(tp):
    traffic = json.load(open(tp))
for name in ("traffic_celt.json", "traffic_mdct.json", "traffic_decode.json", "traffic_silk_frames.json"):
    p = os.path.join(src, name)
    if os.path.exists(p):
        for k, v in json.load(open(p)).items():
            traffic[k] = v
            traffic[k.split("<")[0]] = v
        cp(p, name)
json.dump(traffic, open(tp, "w"), indent=1, sort_keys=True)
pp = os.path.join(src, "pmc_celt.json")
if os.path.exists(pp):
    pmc = {}
    op = os.path.join(ROOT, "profiles", "pmc.json")
    if os.path.exists(op):
        pmc = json.load(open(op))
    pmc.update(json.load(open(pp)))
    json.dump(pmc, open(op, "w"), indent=1, sort_keys=True)
print(sorted(os.listdir(dst)))

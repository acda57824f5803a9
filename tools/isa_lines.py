#!/usr/bin/env python3
"""Static VALU/SALU/LDS instruction counts per source line of a kernel, from `hipcc -S -gline-tables-only`
output (.loc directives). usage: isa_lines.py k.s <kernel-substring> [top]"""
import collections
import re
import sys

path, kern = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 60
files = {}
cur = None
inside = False
cnt = collections.defaultdict(lambda: [0, 0, 0, 0])
for line in open(path):
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', line)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
        continue
    if re.match(r"^_Z\w+:", line):
        inside = kern in line
        continue
    if not inside:
        continue
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", line)
    if m:
        cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
        continue
    m = re.match(r"\s+([a-z_0-9]+)", line)
    if not m or cur is None:
        continue
    op = m.group(1)
    if op.startswith("v_"):
        cnt[cur][0] += 1
    elif op.startswith("s_"):
        cnt[cur][1] += 1
    elif op.startswith("ds_"):
        cnt[cur][2] += 1
    elif op.split("_")[0] in ("global", "scratch", "buffer", "flat"):
        cnt[cur][3] += 1
    if op == "s_endpgm":
        inside = False
tot = [sum(v[i] for v in cnt.values()) for i in range(4)]
print("total valu=%d salu=%d lds=%d vmem=%d" % tuple(tot))
byfile = collections.defaultdict(lambda: [0, 0, 0, 0])
for (f, l), v in cnt.items():
    for i in range(4):
        byfile[f][i] += v[i]
for f, v in sorted(byfile.items(), key=lambda kv: -kv[1][0]):
    print("  %-24s valu=%d salu=%d lds=%d vmem=%d" % (f, *v))
for (f, l), v in sorted(cnt.items(), key=lambda kv: -kv[1][0])[:top]:
    print("%-22s:%-5d valu=%-5d salu=%-5d lds=%-4d vmem=%d" % (f, l, *v))

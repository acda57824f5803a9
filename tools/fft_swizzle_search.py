#!/usr/bin/env python3
"""LDS bank-conflict model of the FFT scratch and the search that produced fsw<0>() (concentus_amd/csrc/mdct_dev.h).

Model (MI355X_MICROARCH.md, LDS table): a wave64 access is served in fixed lane groups, one LDS cycle per group when
conflict-free; every extra distinct address on a busy bank within a group adds a cycle.
    ds_read_b64    2 groups of 32 lanes                    bank = (addr / 4) % 64
    ds_write_b64   4 groups of 16 contiguous lanes         bank = (addr / 4) % 32
    ds_read_b128   4 groups of 16 ({0-3,12-15,20-27}, ...) bank = (addr / 4) % 64
    ds_write_b128  8 groups of 8 contiguous lanes          bank = (addr / 4) % 32

usage:  tools/fft_swizzle_search.py report           conflict cycles per stage, plain layout vs fsw<0>
        tools/fft_swizzle_search.py search [seed]    hill-climb the 15-entry LUT again (minutes)

The stages modelled are those of the 480-point transform in mdct_dev.h: the pre-rotation's scatter (old: one point per lane
at bitrev[i], b64; new: the pair (i, i + 120) per lane, b128), the first butterflies (old: radix 4 at m = 1 and radix 2 at
m = 4 as two passes, b64; new: eight points per lane, b128), radix 4 at m = 8, radix 3 at m = 32, radix 5 at m = 96 and the
linear read-out of the post-rotation."""
import importlib.util
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("gen_tables", os.path.join(HERE, "gen_tables.py"))
gt = importlib.util.module_from_spec(spec)
spec.loader.exec_module(gt)

G_R64 = [list(range(0, 32)), list(range(32, 64))]
G_W64 = [list(range(i, i + 16)) for i in range(0, 64, 16)]
G_R128 = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
G_R128 += [[x + 32 for x in g] for g in G_R128]
G_W128 = [list(range(i, i + 8)) for i in range(0, 64, 8)]
KINDS = {"r64": (8, G_R64, 64), "w64": (8, G_W64, 32), "r128": (16, G_R128, 64), "w128": (16, G_W128, 32)}
LUT = list(gt.FSW_LUT) + [0]                                        # fsw<0> (tools/gen_tables.py is the one place that states it)


def fsw(e, lut=LUT):
    return e ^ (lut[(e >> 5) & 15] << 1) ^ (((e >> 4) & 1) << 2)


def cost(acc, kind, phys):
    """acc: [(lane, point)]; returns (LDS cycles, of which conflict cycles)."""
    width, groups, nbanks = KINDS[kind]
    addr = {l: 8 * phys(e) for l, e in acc}
    tot = base = 0
    for g in groups:
        banks = {}
        for l in g:
            if l in addr:
                for d in range(width // 4):
                    dw = addr[l] // 4 + d
                    banks.setdefault(dw % nbanks, set()).add(dw)
        if banks:
            tot += max(len(s) for s in banks.values())
            base += 1
    return tot, tot - base


def sweeps(cnt, fn, kinds):
    out = []
    for b in range(0, cnt, 64):
        lanes = [l for l in range(64) if b + l < cnt]
        for k in range(len(fn(b))):
            acc = [(l, fn(b + l)[k]) for l in lanes]
            out += [(kd, acc) for kd in kinds]
    return out


def radix(p, m):
    g = 480 // (p * m)

    def f(idx):
        r = idx % (g * m)
        return [(r // m) * p * m + r % m + c * m for c in range(p)]
    return sweeps(480 // p, f, ["r64", "w64"])


def stages(new):
    bitrev = gt.fft_bitrev(480)
    st = {}
    if new:
        sc = []
        for it in range(4):
            acc = []
            for l in range(60):
                n = 60 * it + l
                i = n if n < 120 else n + 120
                assert bitrev[i + 120] == bitrev[i] + 1 and bitrev[i] % 2 == 0
                acc.append((l, bitrev[i]))
            sc.append(("w128", acc))
        st["scatter (pairs, b128)"] = sc
        f8 = []
        for q in range(4):
            acc = [(l, 8 * l + 2 * q) for l in range(60)]
            f8 += [("r128", acc), ("w128", acc)]
        st["first 8 points per lane (b128)"] = f8
    else:
        st["scatter (b64)"] = sweeps(480, lambda i: [bitrev[i % 480]], ["w64"])
        st["radix 4, m 1"] = sweeps(120, lambda i: [4 * i + c for c in range(4)], ["r64", "w64"])
        st["radix 2, m 4"] = sweeps(240, lambda i: [8 * (i >> 2) + (i & 3), 8 * (i >> 2) + (i & 3) + 4], ["r64", "w64"])
    st["radix 4, m 8"] = radix(4, 8)
    st["radix 3, m 32"] = radix(3, 32)
    st["radix 5, m 96"] = radix(5, 96)
    st["read-out"] = sweeps(480, lambda i: [i], ["r64"])
    return st


def total(st, phys):
    t = c = 0
    for ins in st.values():
        for kd, acc in ins:
            a, b = cost(acc, kd, phys)
            t += a
            c += b
    return t, c


def report():
    for name, new, phys in (("plain layout, stages as in round 2", False, lambda e: e), ("fsw<0>, pair scatter, 8 points per lane", True, fsw)):
        st = stages(new)
        print(name)
        for k, ins in st.items():
            t = c = 0
            for kd, acc in ins:
                a, b = cost(acc, kd, phys)
                t += a
                c += b
            print("   %-34s %3d wave-instructions  %4d LDS cycles  %4d of them conflicts" % (k, len(ins), t, c))
        t, c = total(st, phys)
        print("   total %d LDS cycles, %d conflict cycles: conflicts / useful = %.2f" % (t, c, c / (t - c)))


def search(seed):
    rnd = random.Random(seed)
    st = stages(True)
    best = None
    for restart in range(40):
        lut = [rnd.randrange(16) for _ in range(16)]
        cur = total(st, lambda e: fsw(e, lut))[1]
        for _ in range(600):
            t = rnd.randrange(15)
            old = lut[t]
            lut[t] = rnd.randrange(16)
            c = total(st, lambda e: fsw(e, lut))[1]
            if c <= cur:
                cur = c
            else:
                lut[t] = old
        if best is None or cur < best[0]:
            best = (cur, list(lut))
            print("conflict cycles %d  LUT %s" % (cur, " ".join(map(str, lut[:15]))), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "search":
        search(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    else:
        report()

"""Randomised parity sweep on the GPU through the C-ABI: encoder (split pipeline, lane-per-frame back phase) and
decoder against the compiled reference, live, over random settings (see tests/test_fuzz_emu_cpu.py for the same
sweep under host emulation). Skipped where oracle/_ref did not travel."""
import os

import numpy as np
import pytest

import encode_cases as ec
from test_decode_gpu import _gpu_decode
from test_encode_gpu import _gpu_encode

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ca():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "librefdrv.so")):
        pytest.skip("oracle/_ref did not travel")
    import concentus_amd
    concentus_amd.lib.load()
    return concentus_amd


SEEDS = [int(x) for x in os.environ.get("CONCENTUS_FUZZ_SEEDS", "11,22").split(",")]


@pytest.mark.parametrize("seed", SEEDS)
def test_gpu_random_settings_match_reference(ca, seed):
    gm = ec.golden_module()
    rng = np.random.default_rng(seed)
    for _t in range(10):
        br = int(rng.choice([32000, 33000, 34500, 36000, 38400, 40000, 48000, 64000, 96000, 128000, 256000, 510000]))
        vbr, cvbr, cx = int(rng.integers(0, 2)), int(rng.integers(0, 2)), int(rng.integers(0, 11))
        kind, fps, n = str(rng.choice(["noise", "music", "edge"])), int(rng.choice([1, 8, 32])), 128
        pcm = gm.synth_pcm(kind, n, int(rng.integers(1, 1 << 30)))
        if rng.random() < 0.3:
            pcm = (pcm.astype(np.int32) * int(rng.choice([0, 1, 3])) // int(rng.choice([1, 4, 64]))).clip(-32768, 32767).astype(np.int16)
        what = "seed %d: %r" % (seed, dict(br=br, vbr=vbr, cvbr=cvbr, cx=cx, kind=kind, fps=fps))
        pk, ln, rg = gm.ref_encode(gm._Cfg(2, br, vbr, cvbr, cx, 16, 0, 1500), pcm, fps, threads=8)
        out, lens, r2 = _gpu_encode(ca, pcm, fps, (br, vbr, cvbr, cx))
        ec.assert_packets_equal(out, lens, r2, pk, ln, rg, what)
        if (ln > 1).all():
            pkc = np.ascontiguousarray(pk[:, :(int(ln.max()) + 3) & ~3])
            want, wr, wret = gm.ref_decode(pkc, ln, fps, threads=8)
            got, gr, gret = _gpu_decode(ca, pkc, ln.astype(np.int32), fps)
            assert np.array_equal(gret, wret) and np.array_equal(gr, wr) and np.array_equal(got, want), what

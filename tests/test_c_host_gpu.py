"""The outer boundary driven from C (no Python between the caller and the library):

 * the reference's OWN caller, src/opus_demo.c, compiled in place and unedited by oracle/Makefile but linked against
   concentus_amd/compat/libopus.so (the libopus names, forwarding to libopusgpu.so) = oracle/_ref/opus_demo_gpu, next to
   the same source linked against the reference library = oracle/_ref/opus_demo. BASELINE config #1
   (`opus_demo restricted-lowdelay 48000 2 96000`): the .bit files of `-e`, the PCM of `-d` and of the combined
   encode+decode run must be byte-identical (src/opus_demo.c:519-543, :740-830);
 * examples/batch_encode.c, a plain-C host of the batched entry points (gcc -std=c99): one opusgpu_encode_batch +
   one opusgpu_decode_batch over the reference's 48 kHz stereo audio file, against the golden packets.
"""
import os
import struct
import subprocess

import numpy as np
import pytest

import encode_cases as ec

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO_REF = os.path.join(ROOT, "oracle", "_ref", "opus_demo")
DEMO_GPU = os.path.join(ROOT, "oracle", "_ref", "opus_demo_gpu")
BATCH = os.path.join(ROOT, "concentus_amd", "compat", "batch_encode")


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (cmd, r.stdout[-2000:], r.stderr[-2000:])
    return r


@pytest.mark.parametrize("extra", [[], ["-cbr"], ["-cvbr", "-complexity", "5"]], ids=["vbr", "cbr", "cvbr_cx5"])
def test_reference_opus_demo_linked_against_the_gpu_library(tmp_path, extra):
    if not (os.path.exists(DEMO_REF) and os.path.exists(DEMO_GPU)):
        pytest.skip("oracle/_ref/opus_demo(_gpu) did not travel")
    gm = ec.golden_module()
    pcm = np.concatenate([ec.load_case("real48_vbr_stream")[0][100:160], gm.synth_pcm("gmusic", 20, 13371337)])
    pcm_path = tmp_path / "in.pcm"
    pcm.astype("<i2").tofile(pcm_path)
    args = ["restricted-lowdelay", "48000", "2", "96000"] + extra
    out = {}
    for tag, exe in (("ref", DEMO_REF), ("gpu", DEMO_GPU)):
        bit, dec, both = (str(tmp_path / ("%s.%s" % (tag, e))) for e in ("bit", "dec.pcm", "both.pcm"))
        _run([exe, "-e"] + args + [str(pcm_path), bit])
        _run([exe, "-d", "48000", "2", bit, dec])
        _run([exe] + args + [str(pcm_path), both])               # config #1: encode -> decode in one process
        out[tag] = [open(p, "rb").read() for p in (bit, dec, both)]
    assert len(out["ref"][0]) > 80 * 8
    assert out["gpu"][0] == out["ref"][0], ".bit files differ (packets or final ranges)"
    assert out["gpu"][1] == out["ref"][1], "opus_demo -d output differs"
    assert out["gpu"][2] == out["ref"][2], "encode+decode output differs"
    # and crosswise: the reference decodes the GPU build's .bit (its own range check included) to the same PCM
    cross = str(tmp_path / "cross.pcm")
    _run([DEMO_REF, "-d", "48000", "2", str(tmp_path / "gpu.bit"), cross])
    assert open(cross, "rb").read() == out["ref"][1]


def test_plain_c_batch_host_matches_golden_packets(tmp_path):
    if not os.path.exists(BATCH):
        pytest.skip("concentus_amd/compat/batch_encode not built")
    pcm, pk, ln, rg = ec.load_case("real48_vbr_indep")
    pcm_path, bit_path = tmp_path / "real48.pcm", tmp_path / "real48.bit"
    pcm.astype("<i2").tofile(pcm_path)
    r = _run([BATCH, str(pcm_path), str(bit_path)])
    assert "final ranges agree" in r.stderr
    raw = open(bit_path, "rb").read()
    pos = 0
    for k in range(len(ln)):
        n, fr = struct.unpack(">II", raw[pos:pos + 8])
        assert (n, fr) == (int(ln[k]), int(rg[k])), k
        assert raw[pos + 8:pos + 8 + n] == pk[k, :n].tobytes(), k
        pos += 8 + n
    assert pos == len(raw)

"""The opus_demo .bit container (concentus_amd/bitstream.py) against the reference's own command-line tool
(oracle/_ref/opus_demo, compiled in place from opus-fix/src/opus_demo.c by oracle/Makefile): packets this project
holds as golden (byte-identical to what the GPU encoder emits, tests/test_encode_gpu.py) are written as a .bit file,
decoded by the reference CLI -- which also checks every embedded final range against its decoder's -- and the PCM it
writes must be the golden opus_decode() output; and a .bit file the reference CLI encodes is read back and must
carry the golden packets."""
import os
import subprocess

import numpy as np
import pytest

import encode_cases as ec
from concentus_amd import bitstream

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO = os.path.join(ROOT, "oracle", "_ref", "opus_demo")
pytestmark = pytest.mark.skipif(not os.path.exists(DEMO), reason="oracle/_ref/opus_demo not built (needs /root/reference)")


def test_bit_file_decodes_with_reference_cli(tmp_path):
    name = "music_vbr_stream"
    _pcm, pk, ln, rg = ec.load_case(name)
    want = np.load(os.path.join(ROOT, "tests", "golden", "decode_golden.npz"))[name + "_dpcm"]
    bit = tmp_path / "s0.bit"
    out = tmp_path / "s0.pcm"
    bitstream.write_opus_demo_bit(str(bit), pk[:16], ln[:16], rg[:16])          # first stream of the case
    r = subprocess.run([DEMO, "-d", "48000", "2", str(bit), str(out)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    assert "mismatch" not in r.stderr.lower(), r.stderr
    got = np.fromfile(str(out), np.int16).reshape(-1, 2)
    # opus_demo -d drops the decoder delay... it does not for decode-only runs: all 16 x 960 samples are written
    assert got.shape[0] == 16 * 960
    assert np.array_equal(got.reshape(16, 960, 2), want[:16])


def test_reference_cli_bit_file_reads_back_as_golden_packets(tmp_path):
    name = "music_vbr_stream"
    pcm, pk, ln, rg = ec.load_case(name)
    raw = tmp_path / "in.pcm"
    bit = tmp_path / "ref.bit"
    pcm[:16].astype("<i2").tofile(str(raw))
    r = subprocess.run([DEMO, "-e", "restricted-lowdelay", "48000", "2", "96000", str(raw), str(bit)],
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    p2, l2, r2 = bitstream.read_opus_demo_bit(str(bit), stride=1500)
    # the CLI appends one more packet for the zero-padded tail it flushes at end of input (src/opus_demo.c:617-627)
    assert len(l2) == 17
    assert np.array_equal(l2[:16], ln[:16]) and np.array_equal(r2[:16], rg[:16])
    for k in range(16):
        assert np.array_equal(p2[k, :l2[k]], pk[k, :ln[k]])

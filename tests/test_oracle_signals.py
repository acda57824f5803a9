"""Pin oracle/oracle_testsignals.c (generate_music restated) against the reference's own function
(opus-fix/tests/test_opus_encode.c:59-88, compiled in place into oracle/_ref/librefgen.so) and against a committed
digest of its output, so the GPU box (no reference there) generates the identical signal."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

import oraclelib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFGEN = os.path.join(ROOT, "oracle", "_ref", "librefgen.so")
# sha256 of generate_music(96 000 sample pairs) after Rz = Rw = 13371337 and one banner draw (test_opus_encode.c:552-558),
# produced by the reference's own code (librefgen.so) in the build container
DIGEST_96000 = "7dbd497fc291ae1dcb38828b9732f6fadb39fd36ee195aa1c69abe49e94ae8d2"


def gen(n, seed=13371337, skip=1):
    buf = np.zeros((n, 2), np.int16)
    oraclelib.lib().orc_generate_music(oraclelib.ptr(buf), n, C.c_uint32(seed), skip)
    return buf


def test_restated_generate_music_digest():
    assert hashlib.sha256(gen(96000).tobytes()).hexdigest() == DIGEST_96000


@pytest.mark.ref
def test_restated_generate_music_equals_the_reference_function():
    if not os.path.exists(REFGEN):
        pytest.skip("oracle/_ref/librefgen.so not built")
    ref = C.CDLL(REFGEN)
    for n, seed, skip in ((96000, 13371337, 1), (5760, 1, 0), (48000, 0xdeadbeef, 3)):
        want = np.zeros((n, 2), np.int16)
        ref.refgen_music(oraclelib.ptr(want), n, C.c_uint32(seed), skip)
        assert np.array_equal(gen(n, seed, skip), want), (n, seed, skip)
    assert np.abs(gen(96000).astype(np.int32)).max() > 3000          # it is a signal, not silence

"""Pin the CPU restatement of silk_burg_modified / silk_NSQ (oracle/oracle_silk.c) bit-exact against
function-boundary records captured from the compiled reference (tests/golden/silk_golden.npz: 40 records of
synthetic voice + 40 of the real speech file shipped with the reference's Java test console), and -- where
the capture library is present -- against a fresh, larger capture."""
import ctypes as C
import os

import numpy as np
import pytest

import oraclelib

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "silk_golden.npz")


def _check(rec):
    orc = oraclelib.lib()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    bi = np.ascontiguousarray(rec["burg_in"])
    bo = np.zeros_like(rec["burg_out"])
    orc.orc_silk_burg_batch(p(bi), p(bo), bi.shape[0])
    assert np.array_equal(bo, rec["burg_out"]), np.nonzero((bo != rec["burg_out"]).any(1))[0][:8]
    ni = np.ascontiguousarray(rec["nsq_in"])
    st = np.ascontiguousarray(rec["nsq_state_in"]).copy()
    no = np.zeros_like(rec["nsq_out"])
    orc.orc_silk_nsq_batch(p(ni), p(st), p(no), ni.shape[0])
    bad = np.nonzero((no != rec["nsq_out"]).any(1))[0]
    assert bad.size == 0, ("pulses differ in records", bad[:8])
    bad = np.nonzero((st != rec["nsq_state_out"]).any(1))[0]
    assert bad.size == 0, ("NSQ state differs in records", bad[:8])


def test_oracle_silk_matches_golden_records():
    g = np.load(GOLD)
    _check({k[5:]: g[k] for k in g.files})


@pytest.mark.ref
def test_oracle_silk_matches_fresh_capture():
    import encode_cases as ec
    gm = ec.golden_module()
    if not os.path.exists(os.path.join(os.path.dirname(GOLD), "..", "..", "oracle", "_ref", "libopus_ref_silkcap.so")):
        pytest.skip("capture library not built")
    rec = gm.silk_capture(gm.synth_voice(16000 * 3, 99))
    assert rec["nsq_in"].shape[0] >= 100
    _check(rec)


# ---- silk_NSQ_del_dec (opus-fix/silk/NSQ_del_dec.c): 2, 3 and 4 delayed-decision states ----
GOLD_DD = os.path.join(os.path.dirname(GOLD), "silk_dd_golden.npz")


def _check_dd(rec):
    orc = oraclelib.lib()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    di = np.ascontiguousarray(rec["dd_in"])
    st = np.ascontiguousarray(rec["dd_state_in"]).copy()
    do = np.zeros_like(rec["dd_out"])
    orc.orc_silk_nsq_del_dec_batch(p(di), p(st), p(do), di.shape[0])
    nfr = di[:, 8:12].copy().view(np.int32).ravel()                     # frame_length of each record
    for r in range(di.shape[0]):
        assert np.array_equal(do[r, :nfr[r]], rec["dd_out"][r, :nfr[r]]), ("pulses differ in record", r)
        assert np.array_equal(do[r, 320:324], rec["dd_out"][r, 320:324]), ("Seed differs in record", r)
    bad = np.nonzero((st != rec["dd_state_out"]).any(1))[0]
    assert bad.size == 0, ("NSQ state differs in records", bad[:8])


def test_oracle_del_dec_matches_golden_records():
    g = np.load(GOLD_DD)
    rec = {k[5:]: g[k] for k in g.files}
    assert set(rec["dd_in"][:, 1640:1644].copy().view(np.int32).ravel().tolist()) == {2, 3, 4}
    _check_dd(rec)


@pytest.mark.ref
@pytest.mark.parametrize("complexity", [4, 8])
def test_oracle_del_dec_matches_fresh_capture(complexity):
    import encode_cases as ec
    gm = ec.golden_module()
    if not os.path.exists(os.path.join(os.path.dirname(GOLD), "..", "..", "oracle", "_ref", "libopus_ref_silkcap.so")):
        pytest.skip("capture library not built")
    rec = gm.silk_dd_capture(gm.synth_voice(16000 * 3, 77 + complexity), complexity)
    assert rec["dd_in"].shape[0] >= 100
    _check_dd(rec)


def test_oracle_pinned_on_a_fresh_corpus_of_distinct_records():
    """oracle/oracle_silk.c against 4 096 + 3 072 records freshly captured from the unmodified reference encoder
    (tests/silk_corpus.py): Burg, NSQ and NSQ_del_dec outputs and every byte of silk_nsq_state."""
    import ctypes as C
    import silk_corpus
    if not silk_corpus.available():
        pytest.skip("oracle/_ref/libopus_ref_silkcap.so not built")
    import tempfile
    orc = oraclelib.lib()
    p = oraclelib.ptr
    with tempfile.TemporaryDirectory() as tmp:
        c = silk_corpus.corpus(4096, "nsq", cache=tmp, workers=2)
        n = 4096
        bo = np.zeros((n, 72), np.uint8)
        orc.orc_silk_burg_batch(p(np.ascontiguousarray(c["burg_in"])), p(bo), n)
        assert np.array_equal(bo, c["burg_out"])
        st = np.array(c["nsq_state_in"])
        no = np.zeros((n, 320), np.uint8)
        orc.orc_silk_nsq_batch(p(np.ascontiguousarray(c["nsq_in"])), p(st), p(no), n)
        assert np.array_equal(no, c["nsq_out"]) and np.array_equal(st, c["nsq_state_out"])
        d = silk_corpus.corpus(3072, "dd", cache=tmp, workers=2)
        n = 3072
        st = np.array(d["dd_state_in"])
        do = np.zeros((n, 324), np.uint8)
        orc.orc_silk_nsq_del_dec_batch(p(np.ascontiguousarray(d["dd_in"])), p(st), p(do), n)
        assert np.array_equal(do, d["dd_out"]) and np.array_equal(st, d["dd_state_out"])
        del c, d

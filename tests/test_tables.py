"""Generated constant tables (tools/gen_tables.py) against the compiled reference's own tables
(static_modes_fixed.h via the mode struct; file-local statics via .rodata)."""
import pytest

import gen_tables as g
import reflib

pytestmark = pytest.mark.ref


def test_mode_tables_match_reference():
    m = reflib.mode()
    t = g.all_tables()
    A = reflib.arr
    assert t["window120"][1] == A(m.window, 120)
    assert t["eband5ms"][1] == A(m.eBands, 22)
    assert t["logN400"][1] == A(m.logN, 21)
    assert t["band_allocation"][1] == A(m.allocVectors, 231)
    assert t["mdct_trig960"][1] == A(m.mdct.trig, 1800)
    assert t["cache_index50"][1] == A(m.cache.index, 105)
    assert t["cache_bits50"][1] == A(m.cache.bits, 392)
    assert t["cache_caps50"][1] == A(m.cache.caps, 168)
    assert m.cache.size == 392 and m.overlap == 120 and m.nbEBands == 21
    for i, n in enumerate([480, 240, 120, 60]):
        st = m.mdct.kfft[i].contents
        assert st.nfft == n and st.scale == 17476 and st.scale_shift == 8 - i
        assert t["fft_bitrev%d" % n][1] == A(st.bitrev, n)
        fac = [v for pm in g.fft_factors(n) for v in pm]
        assert list(st.factors)[:len(fac)] == fac
    assert t["fft_twiddles480"][1] == A(m.mdct.kfft[0].contents.twiddles, 960)


def test_pvq_table_matches_reference():
    assert reflib.static_table("CELT_PVQ_U_DATA", "I", 1272) == g.pvq_u_table()


def test_emitted_header_is_current(tmp_path):
    import os
    p = tmp_path / "t.h"
    g.emit(str(p))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert p.read_text() == open(os.path.join(root, "concentus_amd", "csrc", "celt_tables.h")).read()
    assert p.read_text() == open(os.path.join(root, "oracle", "oracle_tables.h")).read()


def test_literal_tables_match_reference():
    """Hand-tuned tables listed literally in tools/gen_tables.py vs the reference's .rodata."""
    fmt = {"int8_t": "b", "uint8_t": "B", "int16_t": "h", "uint32_t": "I"}
    ref_name = {"comb_gains": "gains", "tell_frac_correction": "correction", "log2_frac_table": "LOG2_FRAC_TABLE"}
    as_int = {"second_check", "ordery_table"}      # `int` arrays in the reference
    for name, ctype, vals, _cite in g.LITERAL_TABLES:
        if name.endswith("_icdf") and name != "small_energy_icdf":
            f = "B"
        else:
            f = "i" if name in as_int else fmt[ctype]
        got = reflib.static_table(ref_name.get(name, name), f, len(vals))
        assert got == list(vals), name

"""CPU tier: the device sources of the range coder (csrc/rangecoder.h, rangedec.h) driven by coder scripts (csrc/ec_script.h,
host build tests/emu) beside the compiled reference's own ec_enc_* / ec_dec_* functions: celt/tests/test_unit_entropy.c restated
(known-answer vectors for ec_enc_patch_initial_bits, raw-bit overfill, random streams through every symbol method, uniform
integers / raw bits / Laplace symbols)."""
import ctypes as C

import numpy as np
import pytest

import ec_script_cases as ecs
import emulib
import reflib

pytestmark = pytest.mark.skipif(not reflib.available(), reason="oracle/_ref not built")
p = lambda a: a.ctypes.data_as(C.c_void_p)


def emu_enc(e, buf, script):
    st = ecs.pack(e)
    ops = ecs.ops_array(script)
    assert emulib.lib().emu_ec_enc_script(p(st), p(buf), p(ops), len(script)) == 0
    return st


def emu_dec(e, buf, script):
    st = ecs.pack(e)
    ops = ecs.ops_array(script)
    out = np.zeros(len(script), np.int32)
    assert emulib.lib().emu_ec_dec_script(p(st), p(buf), p(ops), len(script), p(out)) == 0
    return st, out


def both_encoders(size, script, run_enc=emu_enc):
    e_ref, b_ref = ecs.fresh_enc(size)
    e_our, b_our = ecs.fresh_enc(size)
    ecs.run_reference(e_ref, script)
    st = run_enc(e_our, b_our, script)
    want = ecs.pack(e_ref)
    assert np.array_equal(st, want), dict(zip(ecs.FIELDS, zip(st.tolist(), want.tolist())))
    assert np.array_equal(b_our[:size], b_ref[:size]), np.nonzero(b_our[:size] != b_ref[:size])[0][:8]
    return e_ref, b_ref


def both_decoders(data, script, run_dec=emu_dec):
    e_ref, b_ref = ecs.fresh_dec(data)
    e_our, b_our = ecs.fresh_dec(data)
    want_out = ecs.run_reference(e_ref, script)
    st, out = run_dec(e_our, b_our, script)
    assert np.array_equal(out, want_out), np.nonzero(out != want_out)[0][:8]
    want = ecs.pack(e_ref)
    assert np.array_equal(st, want), dict(zip(ecs.FIELDS, zip(st.tolist(), want.tolist())))


@pytest.mark.parametrize("case", ecs.known_answer_cases(), ids=lambda c: c[0])
def test_known_answers_of_test_unit_entropy(case):
    name, size, script, expect = case
    e, buf = both_encoders(size, script)
    if "error" in expect:
        assert (e.error != 0) == (expect["error"] != 0), (name, e.error)
    if "range_bytes" in expect:
        assert e.offs == expect["range_bytes"] and buf[0] == expect["byte0"], (name, e.offs, buf[0])


def test_random_streams_through_every_symbol_method():
    for seed in range(40):
        size, enc, dec = ecs.random_stream_case(1000 + seed)
        e, buf = both_encoders(size, enc)
        assert e.error == 0
        both_decoders(buf[:size], dec)


def test_uniform_integers_raw_bits_and_laplace_symbols():
    for seed in range(8):
        size, enc, dec = ecs.uint_bits_case(2000 + seed)
        e, buf = both_encoders(size, enc)
        assert e.error == 0
        both_decoders(buf[:size], dec)


def test_shrink_moves_the_raw_bit_tail():
    script = [(ecs.BITS, 0x1234, 13), (ecs.UINT, 77, 1000), (ecs.BITS, 5, 3), (ecs.SHRINK, 40), (ecs.ENC, 3, 4, 9), (ecs.DONE,)]
    e, buf = both_encoders(200, script)
    assert e.error == 0 and e.storage == 40


def test_script_validation_rejects_what_the_reference_asserts():
    e, buf = ecs.fresh_enc(100)
    st = ecs.pack(e)
    for bad in ([(ecs.ENC, 5, 5, 9)], [(ecs.UINT, 9, 9)], [(ecs.BITS, 8, 3)], [(99, 0, 0, 0)], [(ecs.PATCH, 4, 2)]):
        ops = ecs.ops_array(bad)
        assert emulib.lib().emu_ec_enc_script(p(st), p(buf), p(ops), 1) == -1
    assert np.array_equal(st, ecs.pack(e))

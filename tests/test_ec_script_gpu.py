"""GPU tier: opusgpu_ec_enc_script / opusgpu_ec_dec_script (include/opusgpu_hooks.h) -- the reference's range-coder calls as a
script on the tree's own ec_ctx -- beside the compiled reference's functions: celt/tests/test_unit_entropy.c restated (the
ec_enc_patch_initial_bits vectors :325-358, the raw-bit overfill :359-369, random streams through every symbol method :148-259,
uniform integers / raw bits :71-147, Laplace symbols), every ec_ctx field, every buffer byte and every decoded value compared."""
import ctypes as C

import numpy as np
import pytest

import ec_script_cases as ecs
import reflib
import test_ec_script_cpu as cpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not reflib.available(), reason="oracle/_ref did not travel")]
p = lambda a: a.ctypes.data_as(C.c_void_p)


@pytest.fixture(scope="module")
def L():
    import concentus_amd
    return concentus_amd.lib.load()


def gpu_enc(L):
    def run(e, buf, script):
        ops = ecs.ops_array(script)
        assert L.opusgpu_ec_enc_script(C.byref(e), p(ops), len(script)) == 0, L.opusgpu_get_last_error()
        return ecs.pack(e)
    return run


def gpu_dec(L):
    def run(e, buf, script):
        ops = ecs.ops_array(script)
        out = np.zeros(len(script), np.int32)
        assert L.opusgpu_ec_dec_script(C.byref(e), p(ops), len(script), p(out)) == 0, L.opusgpu_get_last_error()
        return ecs.pack(e), out
    return run


@pytest.mark.parametrize("case", ecs.known_answer_cases(), ids=lambda c: c[0])
def test_known_answers_of_test_unit_entropy(L, case):
    name, size, script, expect = case
    e, buf = cpu.both_encoders(size, script, gpu_enc(L))
    if "error" in expect:
        assert (e.error != 0) == (expect["error"] != 0), (name, e.error)
    if "range_bytes" in expect:
        assert e.offs == expect["range_bytes"] and buf[0] == expect["byte0"], (name, e.offs, buf[0])


def test_random_streams_and_integers_through_the_script_hooks(L):
    for seed in range(12):
        size, enc, dec = ecs.random_stream_case(3000 + seed)
        e, buf = cpu.both_encoders(size, enc, gpu_enc(L))
        assert e.error == 0
        cpu.both_decoders(buf[:size], dec, gpu_dec(L))
    for seed in range(4):
        size, enc, dec = ecs.uint_bits_case(4000 + seed)
        e, buf = cpu.both_encoders(size, enc, gpu_enc(L))
        assert e.error == 0
        cpu.both_decoders(buf[:size], dec, gpu_dec(L))


def test_bad_scripts_run_nothing(L):
    e, buf = ecs.fresh_enc(100)
    before = ecs.pack(e)
    for bad in ([(ecs.ENC, 5, 5, 9)], [(ecs.UINT, 9, 9)], [(99, 0, 0, 0)]):
        ops = ecs.ops_array(bad)
        assert L.opusgpu_ec_enc_script(C.byref(e), p(ops), 1) == -1 and L.opusgpu_get_last_error() == -1
    assert np.array_equal(ecs.pack(e), before)
    big = reflib.EcCtx()
    big.buf = buf.ctypes.data_as(C.POINTER(C.c_ubyte))
    big.storage = 5000
    assert L.opusgpu_ec_enc_script(C.byref(big), p(ecs.ops_array([(ecs.DONE,)])), 1) == -5

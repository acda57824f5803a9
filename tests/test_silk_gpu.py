"""GPU parity tests for the SILK function-level kernels (config #4), through the C-ABI: bit-exact against
records captured from the compiled reference (golden), against the CPU oracle on tiled/perturbed records
at the full 65 536-record size, and against a fresh capture when the capture library travelled."""
import ctypes as C
import os

import numpy as np
import pytest

import oraclelib

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "silk_golden.npz")


@pytest.fixture(scope="module")
def ca():
    import torch
    assert torch.cuda.is_available()
    import concentus_amd
    concentus_amd.lib.load()
    return concentus_amd


def _dev(a):
    import torch
    return torch.from_numpy(np.array(a)).cuda()


def _run_gpu(ca, rec):
    import torch
    bo = ca.silk_burg_modified(_dev(rec["burg_in"]))
    st = _dev(rec["nsq_state_in"])
    pulses = ca.silk_NSQ(_dev(rec["nsq_in"]), st)
    torch.cuda.synchronize()
    return bo.cpu().numpy(), pulses.cpu().numpy().view(np.uint8), st.cpu().numpy()


def test_silk_kernels_match_golden_records(ca):
    g = np.load(GOLD)
    rec = {k[5:]: g[k] for k in g.files}
    bo, pulses, st = _run_gpu(ca, rec)
    assert np.array_equal(bo, rec["burg_out"]), np.nonzero((bo != rec["burg_out"]).any(1))[0][:8]
    assert np.array_equal(pulses, rec["nsq_out"]), np.nonzero((pulses != rec["nsq_out"]).any(1))[0][:8]
    bad = np.nonzero((st != rec["nsq_state_out"]).any(1))[0]
    assert bad.size == 0, ("NSQ state differs", bad[:8], [np.nonzero(st[b] != rec["nsq_state_out"][b])[0][:6] for b in bad[:3]])


def test_silk_kernels_full_size_vs_oracle(ca):
    """65 536 records (config #4): the 80 captured records tiled, with the NSQ dither seed and the Burg
    input perturbed per record so that records differ; checked against the CPU oracle on a 2 048-record sample
    and for batch-position independence."""
    g = np.load(GOLD)
    rec = {k[5:]: g[k] for k in g.files}
    n = 65536
    rng = np.random.default_rng(4)
    reps = n // 80 + 1
    bi = np.tile(rec["burg_in"], (reps, 1))[:n].copy()
    x = bi[:, :768].view(np.int16)
    x += rng.integers(-3, 4, size=x.shape, dtype=np.int16)
    ni = np.tile(rec["nsq_in"], (reps, 1))[:n].copy()
    ni[:, 36:40].view(np.int32)[:, 0] = rng.integers(0, 4, size=n)          # Seed (2 bits in SILK)
    st0 = np.tile(rec["nsq_state_in"], (reps, 1))[:n].copy()
    bo, pulses, st = _run_gpu(ca, dict(burg_in=bi, nsq_in=ni, nsq_state_in=st0))
    idx = rng.choice(n, 2048, replace=False)
    orc = oraclelib.lib()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    sbi = np.ascontiguousarray(bi[idx]); sbo = np.zeros((2048, 72), np.uint8)
    orc.orc_silk_burg_batch(p(sbi), p(sbo), 2048)
    assert np.array_equal(bo[idx], sbo)
    sni = np.ascontiguousarray(ni[idx]); sst = np.ascontiguousarray(st0[idx]).copy(); sno = np.zeros((2048, 320), np.uint8)
    orc.orc_silk_nsq_batch(p(sni), p(sst), p(sno), 2048)
    assert np.array_equal(pulses[idx], sno)
    assert np.array_equal(st[idx], sst)


def test_silk_kernels_match_fresh_capture_if_present(ca):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "oracle", "_ref", "libopus_ref_silkcap.so")):
        pytest.skip("capture library did not travel")
    import encode_cases as ec
    gm = ec.golden_module()
    rec = gm.silk_capture(gm.synth_voice(16000 * 4, 77))
    bo, pulses, st = _run_gpu(ca, rec)
    assert np.array_equal(bo, rec["burg_out"])
    assert np.array_equal(pulses, rec["nsq_out"])
    assert np.array_equal(st, rec["nsq_state_out"])


def test_silk_empty_and_bad_args(ca):
    import torch
    out = ca.silk_burg_modified(torch.zeros((0, 784), dtype=torch.uint8, device="cuda"))
    assert out.shape == (0, 72)
    with pytest.raises(ValueError):
        ca.silk_burg_modified(torch.zeros((4, 100), dtype=torch.uint8, device="cuda"))


# ---- silk_NSQ_del_dec: four lanes per record, one per delayed-decision state ----
GOLD_DD = os.path.join(os.path.dirname(GOLD), "silk_dd_golden.npz")


def _check_dd_gpu(ca, rec, what):
    import torch
    st = _dev(rec["dd_state_in"])
    out = ca.silk_NSQ_del_dec(_dev(rec["dd_in"]), st)
    torch.cuda.synchronize()
    out, st = out.cpu().numpy(), st.cpu().numpy()
    nfr = rec["dd_in"][:, 8:12].copy().view(np.int32).ravel()
    for r in range(out.shape[0]):
        assert np.array_equal(out[r, :nfr[r]], rec["dd_out"][r, :nfr[r]]), (what, "pulses differ in record", r,
                                                                            np.nonzero(out[r, :nfr[r]] != rec["dd_out"][r, :nfr[r]])[0][:8])
        assert np.array_equal(out[r, 320:324], rec["dd_out"][r, 320:324]), (what, "Seed differs in record", r)
    bad = np.nonzero((st != rec["dd_state_out"]).any(1))[0]
    assert bad.size == 0, (what, "NSQ state differs", bad[:8], [np.nonzero(st[b] != rec["dd_state_out"][b])[0][:6] for b in bad[:3]])


def test_del_dec_matches_golden_records(ca):
    g = np.load(GOLD_DD)
    _check_dd_gpu(ca, {k[5:]: g[k] for k in g.files}, "golden")


def test_del_dec_full_size_vs_oracle(ca):
    """65 536 records: the 84 captured ones (2 / 3 / 4 states; unvoiced, voiced, inactive) tiled with the dither seed,
    the number of states and the input perturbed per record, against the CPU oracle on a 2 048-record sample; ragged
    batch sizes (not a multiple of the 16 records of a wavefront)."""
    import torch
    g = np.load(GOLD_DD)
    rec = {k[5:]: g[k] for k in g.files}
    n = 65536 + 7
    rng = np.random.default_rng(9)
    reps = n // 84 + 1
    di = np.tile(rec["dd_in"], (reps, 1))[:n].copy()
    di[:, 36:40].view(np.int32)[:, 0] = rng.integers(0, 4, size=n)            # Seed
    di[:, 1640:1644].view(np.int32)[:, 0] = rng.integers(1, 5, size=n)        # nStatesDelayedDecision 1..4
    x = di[:, 128:128 + 1280].view(np.int32)                                    # x_Q3
    x += rng.integers(-40, 41, size=x.shape, dtype=np.int32)
    st0 = np.tile(rec["dd_state_in"], (reps, 1))[:n].copy()
    st = _dev(st0)
    out = ca.silk_NSQ_del_dec(_dev(di), st)
    torch.cuda.synchronize()
    out, st = out.cpu().numpy(), st.cpu().numpy()
    idx = np.concatenate([rng.choice(n, 2040, replace=False), np.arange(n - 8, n)])
    orc = oraclelib.lib()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    sdi = np.ascontiguousarray(di[idx]); sst = np.ascontiguousarray(st0[idx]).copy(); sdo = np.zeros((idx.size, 324), np.uint8)
    orc.orc_silk_nsq_del_dec_batch(p(sdi), p(sst), p(sdo), idx.size)
    assert np.array_equal(out[idx], sdo), np.nonzero((out[idx] != sdo).any(1))[0][:8]
    assert np.array_equal(st[idx], sst), np.nonzero((st[idx] != sst).any(1))[0][:8]


def test_del_dec_matches_fresh_capture_if_present(ca):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "oracle", "_ref", "libopus_ref_silkcap.so")):
        pytest.skip("capture library did not travel")
    import encode_cases as ec
    gm = ec.golden_module()
    for cx in (4, 6, 9):
        _check_dd_gpu(ca, gm.silk_dd_capture(gm.synth_voice(16000 * 3, 50 + cx), cx), "fresh capture complexity %d" % cx)


def test_burg_per_call_hook_with_reference_signature(ca):
    """opusgpu_silk_burg_modified_c(res_nrg, res_nrg_Q, A_Q16, x, minInvGain_Q30, subfr_length, nb_subfr, D, arch):
    host pointers, the reference's argument list (silk/SigProc_FIX.h:601); against the captured reference outputs."""
    g = np.load(GOLD)
    L = ca.lib.load()
    for r in range(0, 80, 9):
        rec = ca.silk.BurgIn.from_buffer_copy(g["silk_burg_in"][r].tobytes())
        want = ca.silk.BurgOut.from_buffer_copy(g["silk_burg_out"][r].tobytes())
        nrg, nrg_q = C.c_int32(), C.c_int()
        A = (C.c_int32 * 16)()
        x = (C.c_int16 * 384)(*rec.x)
        L.opusgpu_silk_burg_modified_c(C.byref(nrg), C.byref(nrg_q), A, x, rec.minInvGain_Q30, rec.subfr_length, rec.nb_subfr, rec.D, 0)
        assert L.opusgpu_get_last_error() == 0
        assert (nrg.value, nrg_q.value, list(A)[:rec.D]) == (want.res_nrg, want.res_nrg_Q, list(want.A_Q16)[:rec.D]), r
    L.opusgpu_silk_burg_modified_c(C.byref(nrg), C.byref(nrg_q), A, x, 0, 10, 4, 17, 0)      # order > 16
    assert L.opusgpu_get_last_error() == -1


# ---- config #4 at full size on DISTINCT records (tests/silk_corpus.py: every record captured from the unmodified
# reference encoder on its own frame of synthetic speech, with the reference's outputs) ----
def test_silk_kernels_65536_distinct_records_vs_reference_outputs(ca):
    import silk_corpus
    if not silk_corpus.available():
        pytest.skip("capture library did not travel")
    rec = silk_corpus.corpus(65536, "nsq")
    bo, pulses, st = _run_gpu(ca, rec)
    assert np.array_equal(bo, rec["burg_out"]), np.nonzero((bo != rec["burg_out"]).any(1))[0][:8]
    assert np.array_equal(pulses, rec["nsq_out"]), np.nonzero((pulses != rec["nsq_out"]).any(1))[0][:8]
    bad = np.nonzero((st != rec["nsq_state_out"]).any(1))[0]
    assert bad.size == 0, ("NSQ state differs", bad[:8])
    sig = np.asarray(rec["nsq_in"][:, 24:28]).view(np.int32)[:, 0]
    assert (np.bincount(sig, minlength=3) > 2000).all()          # inactive, unvoiced and voiced frames all well represented


def test_del_dec_65536_distinct_records_vs_reference_outputs(ca):
    import torch
    import silk_corpus
    if not silk_corpus.available():
        pytest.skip("capture library did not travel")
    rec = silk_corpus.corpus(65536, "dd")
    st = _dev(rec["dd_state_in"])
    out = ca.silk_NSQ_del_dec(_dev(rec["dd_in"]), st)
    torch.cuda.synchronize()
    out, st = out.cpu().numpy(), st.cpu().numpy()
    assert np.array_equal(out, rec["dd_out"]), np.nonzero((out != rec["dd_out"]).any(1))[0][:8]
    bad = np.nonzero((st != rec["dd_state_out"]).any(1))[0]
    assert bad.size == 0, ("NSQ state differs", bad[:8])
    nst = np.asarray(rec["dd_in"][:, 1640:1644]).view(np.int32)[:, 0]
    assert set(np.unique(nst)) == {2, 3, 4}


def test_corrupted_record_headers_are_skipped_not_executed(ca):
    """The header fields of the records live in device memory; a corrupted one (D > 16, nx > 384, a pitch lag outside
    the LTP memory, 7 delayed-decision states ...) must not index LDS / private arrays out of bounds: the record is
    skipped (outputs zeroed, state untouched) and counted (opusgpu_silk_bad_records), its neighbours are unaffected."""
    import torch
    g = np.load(GOLD)
    rec = {k[5:]: g[k].copy() for k in g.files}
    ca.silk.bad_records()                                        # clear
    bi = rec["burg_in"].copy()
    hdr = bi[:, 768:784].view(np.int32)                         # minInvGain_Q30, subfr_length, nb_subfr, D
    hdr[3, 3] = 17            # D
    hdr[5, 1] = 200           # subfr_length * nb_subfr = 800 > 384
    hdr[7, 2] = 0             # nb_subfr
    hdr[9, 1] = -5
    bo = ca.silk_burg_modified(_dev(bi)).cpu().numpy()
    assert ca.silk.bad_records() == 4
    good = np.setdiff1d(np.arange(80), [3, 5, 7, 9])
    assert np.array_equal(bo[good], rec["burg_out"][good])
    assert (bo[[3, 5, 7, 9], 4:8].view(np.int32) == -2 ** 31).all() and (bo[[3, 5, 7, 9], 8:] == 0).all()
    ni = rec["nsq_in"].copy()
    h = ni[:, :128].view(np.int32)     # nb_subfr, subfr_length, frame_length, ltp_mem_length, predictLPCOrder, shapingLPCOrder, signalType ...
    voiced = np.nonzero(h[:, 6] == 2)[0]
    v = int(voiced[0])
    h[v, 28 + 1] = 100000     # pitchL[1] far outside the LTP memory
    h[2, 0] = 9               # nb_subfr
    h[4, 3] = 4000            # ltp_mem_length
    h[6, 4] = 40              # predictLPCOrder
    bad = sorted({v, 2, 4, 6})
    st = _dev(rec["nsq_state_in"])
    pulses = ca.silk_NSQ(_dev(ni), st)
    torch.cuda.synchronize()
    assert ca.silk.bad_records() == len(bad)
    good = np.setdiff1d(np.arange(80), bad)
    assert np.array_equal(pulses.cpu().numpy().view(np.uint8)[good], rec["nsq_out"][good])
    assert np.array_equal(st.cpu().numpy()[good], rec["nsq_state_out"][good])
    assert np.array_equal(st.cpu().numpy()[bad], rec["nsq_state_in"][bad]) and (pulses.cpu().numpy()[bad] == 0).all()
    gd = np.load(GOLD_DD)
    dd = {k[5:]: gd[k].copy() for k in gd.files}
    di = dd["dd_in"].copy()
    di[1, 1640:1644].view(np.int32)[0] = 7          # nStatesDelayedDecision
    di[3, 4:8].view(np.int32)[0] = 500              # subfr_length
    st = _dev(dd["dd_state_in"])
    out = ca.silk_NSQ_del_dec(_dev(di), st)
    torch.cuda.synchronize()
    assert ca.silk.bad_records() == 2
    good = np.setdiff1d(np.arange(di.shape[0]), [1, 3])
    assert np.array_equal(st.cpu().numpy()[good], dd["dd_state_out"][good])
    assert np.array_equal(st.cpu().numpy()[[1, 3]], dd["dd_state_in"][[1, 3]])
    assert ca.silk.bad_records() == 0


# ---- silk_find_LPC_FIX (first slice): one lane per frame ----
def test_find_lpc_32768_distinct_records_vs_reference_outputs(ca):
    """silk_find_LPC_FIX on the GPU against the outputs the unmodified reference produced when the records were captured
    (tests/silk_corpus.py kind "lpc": complexity 3 / 5 / 8 / 10 in turn, i.e. with and without the NLSF interpolation search)."""
    import torch
    import silk_corpus
    if not silk_corpus.available():
        pytest.skip("capture library did not travel")
    rec = silk_corpus.corpus(32768, "lpc")
    out = ca.silk_find_LPC(_dev(rec["lpc_in"]))
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    want = np.asarray(rec["lpc_out"])
    assert (out[:, 36:40].view(np.int32) == 0).all()
    bad = np.nonzero((out[:, :36] != want[:, :36]).any(1))[0]
    assert bad.size == 0, (bad[:8], out[bad[:2], :36].view(np.int16), want[bad[:2], :36].view(np.int16))
    interp = want[:, 32:36].view(np.int32)[:, 0]
    assert (interp < 4).sum() > 500 and (interp == 4).sum() > 500
    # a corrupted header is skipped and counted, its neighbours are unaffected
    ca.silk.bad_records()
    lin = np.array(rec["lpc_in"][:256])
    lin[7, 768 + 12:768 + 16].view(np.int32)[0] = 40          # predictLPCOrder
    lin[9, 768 + 4:768 + 8].view(np.int32)[0] = 500           # subfr_length
    out2 = ca.silk_find_LPC(_dev(lin)).cpu().numpy()
    assert ca.silk.bad_records() == 2
    good = np.setdiff1d(np.arange(256), [7, 9])
    assert np.array_equal(out2[good, :36], want[:256][good, :36]) and (out2[[7, 9], 36:40].view(np.int32) == -1).all()


# ---- silk_process_NLSFs + silk_residual_energy_FIX (second slice): one lane per record ----
def test_process_nlsfs_and_residual_energy_32768_distinct_records_vs_reference_outputs(ca):
    """The tail of silk_find_pred_coefs_FIX on the GPU against the outputs the unmodified reference produced when the records
    were captured (tests/silk_corpus.py kind "pred": complexity 3 / 5 / 8 / 10 in turn -> different survivor counts, with and
    without NLSF interpolation, voiced and unvoiced): NLSFIndices, quantised NLSFs, both PredCoef_Q12 rows; the four residual
    energies and their Q values."""
    import torch
    import silk_corpus
    if not silk_corpus.available():
        pytest.skip("capture library did not travel")
    rec = silk_corpus.corpus(32768, "pred")
    out = ca.silk_process_NLSFs(_dev(rec["nlsf_in"]))
    eout = ca.silk_residual_energy(_dev(rec["resnrg_in"]))
    torch.cuda.synchronize()
    out, eout = out.cpu().numpy(), eout.cpu().numpy()
    want, ewant = np.asarray(rec["nlsf_out"]), np.asarray(rec["resnrg_out"])
    assert (out[:, 116:120].view(np.int32) == 0).all() and (eout[:, 32:36].view(np.int32) == 0).all()
    bad = np.nonzero((out[:, :116] != want[:, :116]).any(1))[0]
    assert bad.size == 0, (bad.size, bad[:8], out[bad[:1], 96:113].view(np.int8), want[bad[:1], 96:113].view(np.int8))
    bad = np.nonzero((eout[:, :32] != ewant[:, :32]).any(1))[0]
    assert bad.size == 0, (bad.size, bad[:8], eout[bad[:1], :32].view(np.int32), ewant[bad[:1], :32].view(np.int32))
    hdr = np.asarray(rec["nlsf_in"])[:, 64:96].view(np.int32)
    assert len(np.unique(hdr[:, 5])) >= 3 and (hdr[:, 4] < 4).sum() > 500 and (hdr[:, 6] == 2).sum() > 500
    # corrupted headers are skipped and counted, their neighbours are unaffected
    ca.silk.bad_records()
    nin = np.array(rec["nlsf_in"][:256])
    nin[5, 64 + 20:64 + 24].view(np.int32)[0] = 99            # NLSF_MSVQ_Survivors
    nin[11, 64 + 8:64 + 12].view(np.int32)[0] = 12            # predictLPCOrder
    ein = np.array(rec["resnrg_in"][:256])
    ein[3, 848:852].view(np.int32)[0] = 500                   # subfr_length
    o2 = ca.silk_process_NLSFs(_dev(nin)).cpu().numpy()
    e2 = ca.silk_residual_energy(_dev(ein)).cpu().numpy()
    assert ca.silk.bad_records() == 3
    good = np.setdiff1d(np.arange(256), [5, 11])
    assert np.array_equal(o2[good, :116], want[:256][good, :116]) and (o2[[5, 11], 116:120].view(np.int32) == -1).all()
    good = np.setdiff1d(np.arange(256), [3])
    assert np.array_equal(e2[good, :32], ewant[:256][good, :32]) and e2[3, 32:36].view(np.int32)[0] == -1


# ---- silk_find_pred_coefs_FIX, whole (third slice): one lane per frame ----
def test_find_pred_coefs_32768_distinct_records_vs_reference_outputs(ca):
    """silk_find_pred_coefs_FIX on the GPU against everything the unmodified reference wrote when the records were captured
    (tests/silk_corpus.py kind "fpc"; complexity 3 / 5 / 8 / 10 in turn): voiced frames run the LTP analysis, the LTP gain
    codebook search and the LTP residual filter, unvoiced ones the gain-scaled copy; all of them silk_find_LPC_FIX,
    silk_process_NLSFs and silk_residual_energy_FIX."""
    import torch
    import silk_corpus
    if not silk_corpus.available():
        pytest.skip("capture library did not travel")
    rec = silk_corpus.corpus(32768, "fpc")
    out = ca.silk_find_pred_coefs(_dev(rec["fpc_in"]))
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    want = np.asarray(rec["fpc_out"])
    assert (out[:, 204:208].view(np.int32) == 0).all()
    bad = np.nonzero((out[:, :204] != want[:, :204]).any(1))[0]
    assert bad.size == 0, (bad.size, bad[:8], np.nonzero(out[bad[0], :204] != want[bad[0], :204])[0][:12])
    st = np.asarray(rec["fpc_in"])[:, 2624 + 16:2624 + 20].view(np.int32)[:, 0]
    assert (st == 2).sum() > 5000 and (st != 2).sum() > 1000
    # corrupted headers are skipped and counted, their neighbours are unaffected
    ca.silk.bad_records()
    fin = np.array(rec["fpc_in"][:256])
    voiced = np.nonzero(st[:256] == 2)[0]
    fin[voiced[0], 2576:2580].view(np.int32)[0] = 5000        # pitchL[0] of a voiced frame: lag window outside res_pitch
    fin[9, 2624 + 8:2624 + 12].view(np.int32)[0] = 12         # predictLPCOrder
    o2 = ca.silk_find_pred_coefs(_dev(fin)).cpu().numpy()
    assert ca.silk.bad_records() == 2
    good = np.setdiff1d(np.arange(256), [voiced[0], 9])
    assert np.array_equal(o2[good, :204], want[:256][good, :204]) and (o2[[voiced[0], 9], 204:208].view(np.int32) == -1).all()


def test_process_gains_32768_distinct_records_vs_reference_outputs(ca):
    """silk_process_gains_FIX on the GPU against what the unmodified reference wrote when the records were captured
    (tests/silk_corpus.py kind "gains"): quantised and unquantised gains, GainsIndices, LastGainIndex, lastGainIndexPrev,
    quantOffsetType, Lambda_Q10."""
    import torch
    import silk_corpus
    if not silk_corpus.available():
        pytest.skip("capture library did not travel")
    rec = silk_corpus.corpus(32768, "gains")
    out = ca.silk_process_gains(_dev(rec["gains_in"]))
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    want = np.asarray(rec["gains_out"])
    assert (out[:, 52:56].view(np.int32) == 0).all()
    bad = np.nonzero((out[:, :52] != want[:, :52]).any(1))[0]
    assert bad.size == 0, (bad.size, bad[:8], out[bad[:1], :52].view(np.int32), want[bad[:1], :52].view(np.int32))
    ca.silk.bad_records()
    gin = np.array(rec["gains_in"][:256])
    gin[4, 0:4].view(np.int32)[0] = 0                          # Gains_Q16[0]
    gin[8, 48 + 20:48 + 24].view(np.int32)[0] = 200            # LastGainIndex
    o2 = ca.silk_process_gains(_dev(gin)).cpu().numpy()
    assert ca.silk.bad_records() == 2
    good = np.setdiff1d(np.arange(256), [4, 8])
    assert np.array_equal(o2[good, :52], want[:256][good, :52]) and (o2[[4, 8], 52:56].view(np.int32) == -1).all()


def test_noise_shape_analysis_32768_distinct_records_vs_reference_outputs(ca):
    """silk_noise_shape_analysis_FIX on the GPU against what the unmodified reference wrote when the records were captured
    (tests/silk_corpus.py kind "shape"; complexity 3 -> plain autocorrelation, 5 / 8 / 10 -> warped, orders 10 / 12 / 16 / 16)."""
    import torch
    import silk_corpus
    if not silk_corpus.available():
        pytest.skip("capture library did not travel")
    rec = silk_corpus.corpus(32768, "shape")
    out = ca.silk_noise_shape_analysis(_dev(rec["shape_in"]))
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    want = np.asarray(rec["shape_out"])
    assert (out[:, 380:384].view(np.int32) == 0).all()
    bad = np.nonzero((out[:, :380] != want[:, :380]).any(1))[0]
    assert bad.size == 0, (bad.size, bad[:8], np.nonzero(out[bad[0], :380] != want[bad[0], :380])[0][:12])
    hdr = np.asarray(rec["shape_in"])[:, 1600:1696].view(np.int32)
    assert (hdr[:, 6] == 0).sum() > 2000 and (hdr[:, 6] > 0).sum() > 2000
    ca.silk.bad_records()
    sin_ = np.array(rec["shape_in"][:256])
    sin_[6, 1600 + 20:1600 + 24].view(np.int32)[0] = 40          # shapingLPCOrder
    sin_[12, 1600 + 16:1600 + 20].view(np.int32)[0] = 1000       # shapeWinLength
    o2 = ca.silk_noise_shape_analysis(_dev(sin_)).cpu().numpy()
    assert ca.silk.bad_records() == 2
    good = np.setdiff1d(np.arange(256), [6, 12])
    assert np.array_equal(o2[good, :380], want[:256][good, :380]) and (o2[[6, 12], 380:384].view(np.int32) == -1).all()


def test_prefilter_32768_distinct_records_vs_reference_outputs(ca):
    """silk_prefilter_FIX on the GPU against what the unmodified reference produced when the records were captured
    (tests/silk_corpus.py kind "prefilter"): xw_Q3 and every byte of silk_prefilter_state_FIX after the call."""
    import torch
    import silk_corpus
    if not silk_corpus.available():
        pytest.skip("capture library did not travel")
    rec = silk_corpus.corpus(32768, "prefilter")
    st = _dev(rec["prefilter_state_in"])
    out = ca.silk_prefilter(_dev(rec["prefilter_in"]), st)
    torch.cuda.synchronize()
    out, st = out.cpu().numpy(), st.cpu().numpy()
    want, want_st = np.asarray(rec["prefilter_out"]), np.asarray(rec["prefilter_state_out"])
    assert (out[:, 1280:1284].view(np.int32) == 0).all()
    bad = np.nonzero((out[:, :1280] != want[:, :1280]).any(1))[0]
    assert bad.size == 0, (bad.size, bad[:8])
    bad = np.nonzero((st != want_st).any(1))[0]
    assert bad.size == 0, (bad.size, bad[:8], np.nonzero(st[bad[0]] != want_st[bad[0]])[0][:12])
    ca.silk.bad_records()
    xin = np.array(rec["prefilter_in"][:256])
    xin[5, 864 + 20:864 + 24].view(np.int32)[0] = 40              # shapingLPCOrder
    st2 = np.array(rec["prefilter_state_in"][:256])
    st2[9, 1092:1096].view(np.int32)[0] = 4000                   # sLTP_shp_buf_idx
    d_st2 = _dev(st2)
    o2 = ca.silk_prefilter(_dev(xin), d_st2).cpu().numpy()
    assert ca.silk.bad_records() == 2
    good = np.setdiff1d(np.arange(256), [5, 9])
    assert np.array_equal(o2[good, :1280], want[:256][good, :1280]) and (o2[[5, 9], 1280:1284].view(np.int32) == -1).all()
    assert np.array_equal(d_st2.cpu().numpy()[[5, 9]], st2[[5, 9]]), "the state of a skipped record is left as it was"


def test_find_pitch_lags_32768_distinct_records_vs_reference_outputs(ca):
    """silk_find_pitch_lags_FIX (with silk_pitch_analysis_core) on the GPU against what the unmodified reference wrote when the
    records were captured (tests/silk_corpus.py kind "pitch"): the whitened pitch buffer res[], pitchL, lagIndex, contourIndex,
    LTPCorr_Q15, the voicing decision, predGain_Q16."""
    import torch
    import silk_corpus
    if not silk_corpus.available():
        pytest.skip("capture library did not travel")
    rec = silk_corpus.corpus(32768, "pitch")
    out = ca.silk_find_pitch_lags(_dev(rec["pitch_in"]))
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    want = np.asarray(rec["pitch_out"])
    assert (out[:, 1380:1384].view(np.int32) == 0).all()
    bad = np.nonzero((out[:, :1380] != want[:, :1380]).any(1))[0]
    assert bad.size == 0, (bad.size, bad[:8], np.nonzero(out[bad[0], :1380] != want[bad[0], :1380])[0][:12])
    st = want[:, 1372:1376].view(np.int32)[:, 0]
    assert (st == 2).sum() > 5000 and (st == 1).sum() > 1000
    ca.silk.bad_records()
    tin = np.array(rec["pitch_in"][:256])
    tin[3, 1344:1348].view(np.int32)[0] = 12                    # fs_kHz: the 12 kHz path is not provided
    tin[10, 1344 + 24:1344 + 28].view(np.int32)[0] = 40         # pitchEstimationLPCOrder
    o2 = ca.silk_find_pitch_lags(_dev(tin)).cpu().numpy()
    assert ca.silk.bad_records() == 2
    good = np.setdiff1d(np.arange(256), [3, 10])
    assert np.array_equal(o2[good, :1380], want[:256][good, :1380]) and (o2[[3, 10], 1380:1384].view(np.int32) == -1).all()


@pytest.mark.parametrize("variant", ["nb20", "wb10"])
def test_analysis_kernels_at_8_kHz_and_with_10_ms_frames(ca, variant):
    """The six analysis kernels on records captured from the reference at its other operating points (tests/silk_corpus.py
    variants: 8 kHz narrowband input -> order 10, NB/MB codebook, stage-2-only pitch search; 10 ms frames -> nb_subfr 2),
    4 096 records each, every field compared; cf. tests/test_silk_variants_cpu.py."""
    import torch
    import silk_corpus
    if not silk_corpus.available():
        pytest.skip("capture library did not travel")
    ops = (("lpc", "lpc_in", "lpc_out", 36, lambda r: ca.silk_find_LPC(_dev(r["lpc_in"]))),
           ("fpc", "fpc_in", "fpc_out", 204, lambda r: ca.silk_find_pred_coefs(_dev(r["fpc_in"]))),
           ("gains", "gains_in", "gains_out", 52, lambda r: ca.silk_process_gains(_dev(r["gains_in"]))),
           ("shape", "shape_in", "shape_out", 380, lambda r: ca.silk_noise_shape_analysis(_dev(r["shape_in"]))),
           ("pitch", "pitch_in", "pitch_out", 1380, lambda r: ca.silk_find_pitch_lags(_dev(r["pitch_in"]))))
    ca.silk.bad_records()
    for kind, ik, ok, nb, run in ops:
        rec = silk_corpus.corpus(4096, kind, variant=variant, complexities=(4, 9))
        out = run(rec)
        torch.cuda.synchronize()
        out = out.cpu().numpy()
        want = np.asarray(rec[ok])
        bad = np.nonzero((out[:, :nb] != want[:, :nb]).any(1))[0]
        assert bad.size == 0, (kind, variant, bad.size, bad[:6])
    rec = silk_corpus.corpus(4096, "prefilter", variant=variant, complexities=(4, 9))
    st = _dev(rec["prefilter_state_in"])
    out = ca.silk_prefilter(_dev(rec["prefilter_in"]), st).cpu().numpy()
    assert np.array_equal(out[:, :1280], np.asarray(rec["prefilter_out"])[:, :1280])
    assert np.array_equal(st.cpu().numpy(), np.asarray(rec["prefilter_state_out"]))
    assert ca.silk.bad_records() == 0


@pytest.mark.parametrize("variant", ["wb20", "wb40", "nb20", "wb10"])
def test_encode_indices_and_pulses_vs_the_reference_range_coder(ca, variant):
    """silk_encode_indices / silk_encode_pulses on the GPU against the range coder of the unmodified reference captured before and
    after each call (tests/silk_corpus.py kind "bits"): every ec_ctx field and every byte written; then both calls of a frame in
    ONE launch (which = 3) from the coder as it was before silk_encode_indices to the coder after silk_encode_pulses."""
    import torch
    import silk_corpus
    if not silk_corpus.available():
        pytest.skip("capture library did not travel")
    n = 16384 if variant == "wb20" else 4096
    rec = silk_corpus.corpus(n, "bits", variant=variant, complexities=(3, 8))
    ca.silk.bad_records()
    for tag in ("idx", "pls"):
        ec = _dev(rec["bits_%s_ec_in" % tag])
        out = ca.silk_encode_bits(_dev(rec["bits_%s_in" % tag]), ec)
        torch.cuda.synchronize()
        got, want = ec.cpu().numpy(), np.asarray(rec["bits_%s_ec_out" % tag])
        bad = np.nonzero((got != want).any(1))[0]
        assert bad.size == 0, (tag, variant, bad.size, bad[:6], np.nonzero(got[bad[0]] != want[bad[0]])[0][:12])
        if tag == "idx":
            assert np.array_equal(out.cpu().numpy()[:, :8], np.asarray(rec["bits_idx_out"])[:, :8])
    both = np.array(rec["bits_idx_in"])
    both[:, :320] = np.asarray(rec["bits_pls_in"])[:, :320]                      # the frame's pulses
    both[:, 348 + 60:348 + 64].view(np.int32)[:, 0] = 3                           # which
    ec = _dev(rec["bits_idx_ec_in"])
    ca.silk_encode_bits(_dev(both), ec)
    assert np.array_equal(ec.cpu().numpy(), np.asarray(rec["bits_pls_ec_out"]))
    assert ca.silk.bad_records() == 0
    bad_in = np.array(rec["bits_idx_in"][:64])
    bad_in[5, 320] = 99                                                           # GainsIndices[0] outside its model
    ec2 = _dev(rec["bits_idx_ec_in"][:64])
    o2 = ca.silk_encode_bits(_dev(bad_in), ec2).cpu().numpy()
    assert ca.silk.bad_records() == 1 and o2[5, 8:12].view(np.int32)[0] == -1
    assert np.array_equal(ec2.cpu().numpy()[5], np.asarray(rec["bits_idx_ec_in"])[5]), "a skipped record leaves its coder untouched"


@pytest.mark.parametrize("variant", ["wb20", "nb20", "wb10"])
def test_vad_vs_reference_outputs(ca, variant):
    """silk_VAD_GetSA_Q8_c on the GPU against the unmodified reference (tests/silk_corpus.py kind "vad", from the first frame of every
    encoder so that the start-up of the noise-level tracker is covered): speech_activity_Q8, input_tilt_Q15,
    input_quality_bands_Q15[4] and every byte of silk_VAD_state after the call."""
    import torch
    import silk_corpus
    if not silk_corpus.available():
        pytest.skip("capture library did not travel")
    rec = silk_corpus.corpus(16384 if variant == "wb20" else 4096, "vad", variant=variant, complexities=(5,))
    st = _dev(rec["vad_state_in"])
    out = ca.silk_VAD_GetSA_Q8(_dev(rec["vad_in"]), st)
    torch.cuda.synchronize()
    out, st = out.cpu().numpy(), st.cpu().numpy()
    assert np.array_equal(out[:, :24], np.asarray(rec["vad_out"])[:, :24]) and (out[:, 24:28].view(np.int32) == 0).all()
    bad = np.nonzero((st != np.asarray(rec["vad_state_out"])).any(1))[0]
    assert bad.size == 0, (bad.size, bad[:6])
    sa = np.asarray(rec["vad_out"])[:, :4].view(np.int32)[:, 0]
    assert (sa > 200).sum() > 500 and (sa < 30).sum() > 200, "active speech and pauses"

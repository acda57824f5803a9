"""GPU tier: the per-call hooks of include/opusgpu_hooks.h, called with the reference's own argument lists, against the
SAME functions of the compiled reference (oracle/_ref/libopus_ref.so; the two that are static in the reference through
the trampolines of oracle/_ref/librefhooks.so): opus_ifft, comb_filter_const, exp_rotation1, renormalise_vector,
silk_NSQ, silk_NSQ_del_dec. Bit-exact, including the in-place / state side effects."""
import ctypes as C
import os

import numpy as np
import pytest

import reflib
from test_hooks_layout import header_defines

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFHOOKS = os.path.join(ROOT, "oracle", "_ref", "librefhooks.so")


@pytest.fixture(scope="module")
def L():
    import torch
    assert torch.cuda.is_available()
    import concentus_amd
    return concentus_amd.lib.load()


@pytest.fixture(scope="module")
def ref():
    if not (reflib.available() and os.path.exists(REFHOOKS)):
        pytest.skip("oracle/_ref did not travel")
    return reflib.lib(), C.CDLL(REFHOOKS)


def p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_opus_ifft(L, ref):
    r, _ = ref
    m = reflib.mode()
    rng = np.random.default_rng(21)
    for k in range(4):
        st = m.mdct.kfft[k]
        n = 480 >> k
        fin = rng.integers(-(1 << 24), 1 << 24, size=(n, 2), dtype=np.int32)
        want = np.zeros((n, 2), np.int32)
        r.opus_ifft_c(st, p(fin), p(want))
        got = np.zeros((n, 2), np.int32)
        L.opusgpu_opus_ifft(st, p(fin), p(got))
        assert L.opusgpu_get_last_error() == 0
        assert np.array_equal(got, want), k
    L.opusgpu_opus_ifft(m.mdct.kfft[0], p(fin), p(fin))                # in place: rejected like the reference's assert
    assert L.opusgpu_get_last_error() == -1


def test_comb_filter_const(L, ref):
    _, h = ref
    rng = np.random.default_rng(22)
    for T, N, g in ((15, 960, (9830, 7000, 4200)), (1022, 960, (-15000, 8784, 0)), (300, 840, (26208, 3280, 0)), (77, 5, (1, 2, 3))):
        buf = rng.integers(-(1 << 27), 1 << 27, size=T + 2 + N + 8, dtype=np.int32)
        # separate output (the encoder's pre-filter): pure FIR
        y0 = np.zeros(N, np.int32)
        x = buf.copy()
        h.refhook_comb_filter_const(p(y0), C.c_void_p(x.ctypes.data + 4 * (T + 2)), T, N, *g)
        y1 = np.zeros(N, np.int32)
        x1 = buf.copy()
        L.opusgpu_comb_filter_const(p(y1), C.c_void_p(x1.ctypes.data + 4 * (T + 2)), T, N, *g)
        assert L.opusgpu_get_last_error() == 0
        assert np.array_equal(y1, y0) and np.array_equal(x1, buf), (T, N)
        # in place (the decoder's post-filter): recursive through the samples already filtered
        a, b = buf.copy(), buf.copy()
        pa, pb = C.c_void_p(a.ctypes.data + 4 * (T + 2)), C.c_void_p(b.ctypes.data + 4 * (T + 2))
        h.refhook_comb_filter_const(pa, pa, T, N, *g)
        L.opusgpu_comb_filter_const(pb, pb, T, N, *g)
        assert L.opusgpu_get_last_error() == 0
        assert np.array_equal(a, b), (T, N, "in place")
    a = buf.copy()
    L.opusgpu_comb_filter_const(C.c_void_p(a.ctypes.data + 4 * 10), C.c_void_p(a.ctypes.data + 4 * 100), 50, 60, 1, 2, 3)
    assert L.opusgpu_get_last_error() == -1                             # partially overlapping spans


def test_exp_rotation1(L, ref):
    _, h = ref
    rng = np.random.default_rng(23)
    for n, stride in ((8, 1), (16, 1), (24, 3), (48, 1), (96, 4), (176, 1), (176, 13), (144, 8), (5, 4), (3, 1), (64, 63)):
        X = rng.integers(-16384, 16384, size=n, dtype=np.int16)
        c, s = int(rng.integers(1, 32767)), int(rng.integers(-32767, 32767))
        a, b = X.copy(), X.copy()
        h.refhook_exp_rotation1(p(a), n, stride, c, s)
        L.opusgpu_exp_rotation1(p(b), n, stride, c, s)
        assert L.opusgpu_get_last_error() == 0
        assert np.array_equal(a, b), (n, stride)


def test_renormalise_vector(L, ref):
    r, _ = ref
    rng = np.random.default_rng(24)
    for n in (2, 8, 13, 24, 96, 176, 960):
        for amp, gain in ((16000, 32767), (300, 32767), (16000, 23170), (3, 12345)):
            X = rng.integers(-amp, amp + 1, size=n, dtype=np.int16)
            a, b = X.copy(), X.copy()
            r.renormalise_vector(p(a), n, gain, 0)
            L.opusgpu_renormalise_vector(p(b), n, gain, 0)
            assert L.opusgpu_get_last_error() == 0
            assert np.array_equal(a, b), (n, amp, gain)


def _ref_structs(rec_in, dd):
    """The reference's silk_encoder_state / SideInfoIndices (zeroed, only the fields the quantisers read are set, at the
    offsets of include/opusgpu_hooks.h) from one function-boundary record."""
    d = header_defines()
    hdr = rec_in[:48].view(np.int32)          # nb_subfr, subfr_length, frame_length, ltp_mem_length, predictLPCOrder, shapingLPCOrder, signalType, quantOffsetType, NLSFInterpCoef_Q2, Seed, Lambda_Q10, LTP_scale_Q14
    enc = np.zeros(d["OPUSGPU_REF_SIZEOF_SILK_ENCODER_STATE"], np.uint8)
    for off, v in ((d["OPUSGPU_REF_OFF_NB_SUBFR"], hdr[0]), (d["OPUSGPU_REF_OFF_SUBFR_LENGTH"], hdr[1]),
                   (d["OPUSGPU_REF_OFF_FRAME_LENGTH"], hdr[2]), (d["OPUSGPU_REF_OFF_LTP_MEM_LENGTH"], hdr[3]),
                   (d["OPUSGPU_REF_OFF_PREDICT_LPC_ORDER"], hdr[4]), (d["OPUSGPU_REF_OFF_SHAPING_LPC_ORDER"], hdr[5])):
        enc[off:off + 4].view(np.int32)[0] = v
    if dd is not None:
        enc[d["OPUSGPU_REF_OFF_N_STATES_DEL_DEC"]:][:4].view(np.int32)[0] = dd[0]
        enc[d["OPUSGPU_REF_OFF_WARPING_Q16"]:][:4].view(np.int32)[0] = dd[1]
    idx = np.zeros(d["OPUSGPU_REF_SIZEOF_SIDE_INFO_INDICES"], np.int8)
    idx[d["OPUSGPU_REF_OFF_SIGNAL_TYPE"]] = hdr[6]
    idx[d["OPUSGPU_REF_OFF_QUANT_OFFSET_TYPE"]] = hdr[7]
    idx[d["OPUSGPU_REF_OFF_NLSF_INTERP_COEF_Q2"]] = hdr[8]
    idx[d["OPUSGPU_REF_OFF_SEED"]] = hdr[9]
    return enc, idx, int(hdr[10]), int(hdr[11])


def _arrays(rec_in):
    i32 = lambda lo, n: np.ascontiguousarray(rec_in[lo:lo + 4 * n].view(np.int32))
    harm, tilt, lf, gains, pitch = (i32(48 + 16 * k, 4) for k in range(5))
    x_Q3 = i32(128, 320)
    i16 = lambda lo, n: np.ascontiguousarray(rec_in[lo:lo + 2 * n].view(np.int16))
    return x_Q3, i16(1408, 32), i16(1472, 20), i16(1512, 64), harm, tilt, lf, gains, pitch


@pytest.mark.parametrize("which", ["nsq", "del_dec"])
def test_silk_nsq_hooks_with_the_reference_argument_list(L, ref, which):
    r, _ = ref
    gold = np.load(os.path.join(ROOT, "tests", "golden", "silk_golden.npz" if which == "nsq" else "silk_dd_golden.npz"))
    key = "silk_nsq_" if which == "nsq" else "silk_dd_"
    rin = gold["silk_nsq_in" if which == "nsq" else "silk_dd_in"]
    st_in = gold[key + "state_in"]
    fn_ref = r.silk_NSQ_c if which == "nsq" else r.silk_NSQ_del_dec_c
    fn_gpu = L.opusgpu_silk_NSQ if which == "nsq" else L.opusgpu_silk_NSQ_del_dec
    for k in range(0, rin.shape[0], 5):
        rec = np.ascontiguousarray(rin[k])
        dd = None if which == "nsq" else rec[1640:1648].view(np.int32)
        x_Q3, pred, ltp, ar2, harm, tilt, lf, gains, pitch = _arrays(rec)
        out = []
        for fn in (fn_ref, fn_gpu):
            enc, idx, lam, ltps = _ref_structs(rec, dd)
            st = np.ascontiguousarray(st_in[k]).copy()
            pulses = np.zeros(320, np.int8)
            fn(p(enc), p(st), p(idx), p(x_Q3), p(pulses), p(pred), p(ltp), p(ar2), p(harm), p(tilt), p(lf), p(gains), p(pitch),
               C.c_int(lam), C.c_int(ltps))
            out.append((st, pulses, idx))
        assert L.opusgpu_get_last_error() == 0
        assert np.array_equal(out[0][1], out[1][1]), (which, k, "pulses")
        assert np.array_equal(out[0][0], out[1][0]), (which, k, "silk_nsq_state")
        assert np.array_equal(out[0][2], out[1][2]), (which, k, "SideInfoIndices (Seed)")
        if which == "nsq":
            assert np.array_equal(out[1][1].view(np.uint8), gold["silk_nsq_out"][k])


def test_silk_find_LPC_FIX_hook_with_the_reference_argument_list(L, ref):
    """opusgpu_silk_find_LPC_FIX(psEncC, NLSF_Q15, x, minInvGain_Q30) against silk_find_LPC_FIX of the compiled reference, both
    driven through a zeroed silk_encoder_state in which only the fields the function reads are set (offsets of
    include/opusgpu_hooks.h); compared: NLSF_Q15[] and psEncC->indices.NLSFInterpCoef_Q2."""
    import silk_corpus
    r, _ = ref
    if not silk_corpus.available():
        pytest.skip("capture library did not travel")
    d = header_defines()
    rec = silk_corpus.corpus(4096, "lpc")
    lin = np.asarray(rec["lpc_in"])
    want = np.asarray(rec["lpc_out"])
    picks = list(range(0, 4096, 293))
    for k in picks:
        row = np.ascontiguousarray(lin[k])
        hdr = row[768:792].view(np.int32)
        x = np.ascontiguousarray(row[:768].view(np.int16))
        outs = []
        for fn in (r.silk_find_LPC_FIX, L.opusgpu_silk_find_LPC_FIX):
            enc = np.zeros(d["OPUSGPU_REF_SIZEOF_SILK_ENCODER_STATE"], np.uint8)
            for off, v in ((d["OPUSGPU_REF_OFF_SUBFR_LENGTH"], hdr[1]), (d["OPUSGPU_REF_OFF_NB_SUBFR"], hdr[2]),
                           (d["OPUSGPU_REF_OFF_PREDICT_LPC_ORDER"], hdr[3]), (d["OPUSGPU_REF_OFF_USE_INTERPOLATED_NLSFS"], hdr[4]),
                           (d["OPUSGPU_REF_OFF_FIRST_FRAME_AFTER_RESET"], hdr[5])):
                enc[off:off + 4].view(np.int32)[0] = v
            enc[d["OPUSGPU_REF_OFF_PREV_NLSFQ_Q15"]:][:32] = row[792:824]
            nlsf = np.zeros(16, np.int16)
            fn(p(enc), p(nlsf), p(x), C.c_int32(int(hdr[0])))
            outs.append((nlsf, int(enc[d["OPUSGPU_REF_OFF_INDICES"] + d["OPUSGPU_REF_OFF_NLSF_INTERP_COEF_Q2"]].view(np.int8))))
        assert L.opusgpu_get_last_error() == 0
        assert np.array_equal(outs[0][0], outs[1][0]) and outs[0][1] == outs[1][1], k
        assert np.array_equal(outs[1][0], want[k, :32].view(np.int16)) and outs[1][1] == want[k, 32:36].view(np.int32)[0]


def test_silk_process_NLSFs_and_residual_energy_hooks_with_the_reference_argument_lists(L, ref):
    """opusgpu_silk_process_NLSFs(psEncC, PredCoef_Q12, pNLSF_Q15, prev_NLSFq_Q15) and opusgpu_silk_residual_energy_FIX(nrgs, nrgsQ,
    x, a_Q12, gains, subfr_length, nb_subfr, LPC_order, arch) against the same functions of the compiled reference. The first
    is driven through a zeroed silk_encoder_state in which only the fields it reads are set (offsets of
    include/opusgpu_hooks.h) -- except psNLSF_CB, a pointer into the reference's own tables, which the reference run needs and
    takes from a silk_encoder_state the reference initialised itself."""
    import silk_corpus
    r, _ = ref
    if not silk_corpus.available():
        pytest.skip("capture library did not travel")
    d = header_defines()
    rec = silk_corpus.corpus(4096, "pred")
    nin, nwant = np.asarray(rec["nlsf_in"]), np.asarray(rec["nlsf_out"])
    ein, ewant = np.asarray(rec["resnrg_in"]), np.asarray(rec["resnrg_out"])
    # psNLSF_CB: offset from oracle/_ref/layout.json (offsetof() compiled against the reference's headers); the GPU hook never reads it
    import json
    off_cb = json.load(open(os.path.join(ROOT, "oracle", "_ref", "layout.json")))["silk_encoder_state.psNLSF_CB"]
    cb_wb = C.addressof(C.c_char.in_dll(r, "silk_NLSF_CB_WB"))
    cb_nb = C.addressof(C.c_char.in_dll(r, "silk_NLSF_CB_NB_MB"))
    for k in range(0, 4096, 293):
        row = np.ascontiguousarray(nin[k])
        hdr = row[64:96].view(np.int32)
        outs = []
        for which, fn in (("ref", r.silk_process_NLSFs), ("gpu", L.opusgpu_silk_process_NLSFs)):
            enc = np.zeros(d["OPUSGPU_REF_SIZEOF_SILK_ENCODER_STATE"] + 64, np.uint8)
            for off, v in ((d["OPUSGPU_REF_OFF_SPEECH_ACTIVITY_Q8"], hdr[0]), (d["OPUSGPU_REF_OFF_NB_SUBFR"], hdr[1]),
                           (d["OPUSGPU_REF_OFF_PREDICT_LPC_ORDER"], hdr[2]), (d["OPUSGPU_REF_OFF_USE_INTERPOLATED_NLSFS"], hdr[3]),
                           (d["OPUSGPU_REF_OFF_NLSF_MSVQ_SURVIVORS"], hdr[5])):
                enc[off:off + 4].view(np.int32)[0] = v
            ind = d["OPUSGPU_REF_OFF_INDICES"]
            enc[ind + d["OPUSGPU_REF_OFF_NLSF_INTERP_COEF_Q2"]] = np.uint8(hdr[4])
            enc[ind + d["OPUSGPU_REF_OFF_SIGNAL_TYPE"]] = np.uint8(hdr[6])
            if which == "ref":
                enc[off_cb:off_cb + 8].view(np.uint64)[0] = cb_wb if hdr[2] == 16 else cb_nb
            pc = np.zeros(32, np.int16)
            nlsf = np.ascontiguousarray(row[:32].view(np.int16)).copy()
            prev = np.ascontiguousarray(row[32:64].view(np.int16)).copy()
            fn(p(enc), p(pc), p(nlsf), p(prev))
            outs.append((pc, nlsf, enc[ind + d["OPUSGPU_REF_OFF_NLSF_INDICES"]:][:17].copy()))
        assert L.opusgpu_get_last_error() == 0
        for a, b in zip(outs[0], outs[1]):
            assert np.array_equal(a, b), k
        assert np.array_equal(outs[1][0].view(np.uint8), nwant[k, :64]) and np.array_equal(outs[1][2], nwant[k, 96:113])

        row = np.ascontiguousarray(ein[k])
        eh = row[848:864].view(np.int32)
        outs = []
        for fn in (r.silk_residual_energy_FIX, L.opusgpu_silk_residual_energy_FIX):
            nrgs, nrgsQ = np.zeros(4, np.int32), np.zeros(4, np.int32)
            x = np.ascontiguousarray(row[:768].view(np.int16)).copy()
            a = np.ascontiguousarray(row[768:832].view(np.int16)).copy()
            g = np.ascontiguousarray(row[832:848].view(np.int32)).copy()
            fn(p(nrgs), p(nrgsQ), p(x), p(a), p(g), C.c_int(int(eh[0])), C.c_int(int(eh[1])), C.c_int(int(eh[2])), C.c_int(0))
            outs.append((nrgs, nrgsQ))
        assert L.opusgpu_get_last_error() == 0
        assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1]), k
        assert np.array_equal(outs[1][0].view(np.uint8), ewant[k, :16]) and np.array_equal(outs[1][1].view(np.uint8), ewant[k, 16:32])


def test_silk_find_pred_coefs_FIX_hook_with_the_reference_argument_list(L, ref):
    """opusgpu_silk_find_pred_coefs_FIX(psEnc, psEncCtrl, res_pitch, x, condCoding) against silk_find_pred_coefs_FIX of the compiled
    reference: both are handed the same silk_encoder_state_FIX / silk_encoder_control_FIX images (zero except the fields the
    function reads, at the offsets of include/opusgpu_hooks.h, plus psNLSF_CB for the reference) and the same res_pitch / x
    buffers; compared: EVERY byte of both structures after the call, i.e. all the fields written and nothing else touched."""
    import json
    import silk_corpus
    r, _ = ref
    if not silk_corpus.available():
        pytest.skip("capture library did not travel")
    d = header_defines()
    off_cb = json.load(open(os.path.join(ROOT, "oracle", "_ref", "layout.json")))["silk_encoder_state.psNLSF_CB"]
    cb_wb = C.addressof(C.c_char.in_dll(r, "silk_NLSF_CB_WB"))
    rec = silk_corpus.corpus(4096, "fpc")
    fin, want = np.asarray(rec["fpc_in"]), np.asarray(rec["fpc_out"])
    names = ("nb_subfr", "subfr_length", "predictLPCOrder", "ltp_mem_length", "signalType", "condCoding", "first_frame_after_reset",
             "useInterpolatedNLSFs", "speech_activity_Q8", "NLSF_MSVQ_Survivors", "mu_LTP_Q9", "LTPQuantLowComplexity",
             "sum_log_gain_Q7", "coding_quality_Q14", "PacketLoss_perc", "nFramesPerPacket")
    state_off = {"nb_subfr": "NB_SUBFR", "subfr_length": "SUBFR_LENGTH", "predictLPCOrder": "PREDICT_LPC_ORDER",
                 "ltp_mem_length": "LTP_MEM_LENGTH", "first_frame_after_reset": "FIRST_FRAME_AFTER_RESET",
                 "useInterpolatedNLSFs": "USE_INTERPOLATED_NLSFS", "speech_activity_Q8": "SPEECH_ACTIVITY_Q8",
                 "NLSF_MSVQ_Survivors": "NLSF_MSVQ_SURVIVORS", "mu_LTP_Q9": "MU_LTP_Q9", "LTPQuantLowComplexity": "LTP_QUANT_LOW_COMPLEXITY",
                 "sum_log_gain_Q7": "SUM_LOG_GAIN_Q7", "PacketLoss_perc": "PACKET_LOSS_PERC", "nFramesPerPacket": "N_FRAMES_PER_PACKET"}
    seen = set()
    for k in range(0, 4096, 157):
        row = np.ascontiguousarray(fin[k])
        hdr = dict(zip(names, row[2624:2688].view(np.int32).tolist()))
        seen.add(hdr["signalType"] == 2)
        images = []
        for fn in (r.silk_find_pred_coefs_FIX, L.opusgpu_silk_find_pred_coefs_FIX):
            enc = np.zeros(32768, np.uint8)                   # silk_encoder_state_FIX starts with sCmn; the rest is never touched
            ctl = np.zeros(d["OPUSGPU_REF_SIZEOF_SILK_ENCODER_CONTROL_FIX"], np.uint8)
            for name, macro in state_off.items():
                o = d["OPUSGPU_REF_OFF_" + macro]
                enc[o:o + 4].view(np.int32)[0] = hdr[name]
            # frame_length is nb_subfr * subfr_length (silk/control_codec.c); the reference reads it for its stack buffer
            o = d["OPUSGPU_REF_OFF_FRAME_LENGTH"]
            enc[o:o + 4].view(np.int32)[0] = hdr["nb_subfr"] * hdr["subfr_length"]
            enc[d["OPUSGPU_REF_OFF_INDICES"] + d["OPUSGPU_REF_OFF_SIGNAL_TYPE"]] = np.uint8(hdr["signalType"])
            enc[d["OPUSGPU_REF_OFF_PREV_NLSFQ_Q15"]:][:32] = row[2592:2624]
            enc[off_cb:off_cb + 8].view(np.uint64)[0] = cb_wb
            ctl[d["OPUSGPU_REF_OFF_CTRL_GAINS_Q16"]:][:16] = row[2560:2576]
            ctl[d["OPUSGPU_REF_OFF_CTRL_PITCHL"]:][:16] = row[2576:2592]
            o = d["OPUSGPU_REF_OFF_CTRL_CODING_QUALITY_Q14"]
            ctl[o:o + 4].view(np.int32)[0] = hdr["coding_quality_Q14"]
            res_pitch = np.zeros(1024, np.int16)
            res_pitch[:640] = row[:1280].view(np.int16)
            xbuf = np.zeros(1024, np.int16)
            xbuf[:640] = row[1280:2560].view(np.int16)
            xptr = C.c_void_p(xbuf.ctypes.data + 2 * hdr["ltp_mem_length"])
            fn(p(enc), p(ctl), p(res_pitch), xptr, C.c_int(hdr["condCoding"]))
            images.append((enc, ctl))
        assert L.opusgpu_get_last_error() == 0
        assert np.array_equal(images[0][1], images[1][1]), (k, "silk_encoder_control_FIX", np.nonzero(images[0][1] != images[1][1])[0][:8])
        assert np.array_equal(images[0][0], images[1][0]), (k, "silk_encoder_state_FIX", np.nonzero(images[0][0] != images[1][0])[0][:8])
        ctl = images[1][1]
        assert np.array_equal(ctl[d["OPUSGPU_REF_OFF_CTRL_PRED_COEF_Q12"]:][:64], want[k, :64])
    assert seen == {True, False}, "voiced and unvoiced frames among the picks"


def _fix_images(d, sets_state, sets_ctrl):
    """A zeroed silk_encoder_state_FIX / silk_encoder_control_FIX pair with the given (macro suffix, value, ctype) fields set."""
    enc = np.zeros(d["OPUSGPU_REF_SIZEOF_SILK_ENCODER_STATE_FIX"] + 64, np.uint8)
    ctl = np.zeros(d["OPUSGPU_REF_SIZEOF_SILK_ENCODER_CONTROL_FIX"], np.uint8)
    for img, sets in ((enc, sets_state), (ctl, sets_ctrl)):
        for off, val, dt in sets:
            a = np.atleast_1d(np.asarray(val)).astype(dt)
            img[off:off + a.nbytes] = a.view(np.uint8)
    return enc, ctl


def test_the_four_frame_analysis_hooks_with_the_reference_argument_lists(L, ref):
    """opusgpu_silk_find_pitch_lags_FIX / _noise_shape_analysis_FIX / _process_gains_FIX / _prefilter_FIX against the same functions
    of the compiled reference: both get identical silk_encoder_state_FIX / silk_encoder_control_FIX images (zero except the fields
    the function reads, placed at the offsets of include/opusgpu_hooks.h) and identical signal buffers; compared afterwards:
    EVERY byte of both structures and the output arrays (res[], xw_Q3[])."""
    import silk_corpus
    from concentus_amd import silk as S
    r, _ = ref
    if not silk_corpus.available():
        pytest.skip("capture library did not travel")
    d = header_defines()
    O = lambda name: d["OPUSGPU_REF_OFF_" + name]
    IND, SHP, PF = O("INDICES"), O("FIX_SSHAPE"), O("FIX_SPREFILT")
    i32, i8, i16 = np.int32, np.int8, np.int16

    def both(fn_ref, fn_gpu, make_args):
        imgs = []
        for fn in (fn_ref, fn_gpu):
            enc, ctl, extra, call = make_args()
            call(fn, enc, ctl, extra)
            imgs.append((enc, ctl, extra))
        assert L.opusgpu_get_last_error() == 0
        assert np.array_equal(imgs[0][1], imgs[1][1]), ("silk_encoder_control_FIX", np.nonzero(imgs[0][1] != imgs[1][1])[0][:8])
        assert np.array_equal(imgs[0][0], imgs[1][0]), ("silk_encoder_state_FIX", np.nonzero(imgs[0][0] != imgs[1][0])[0][:8])
        for a, b in zip(imgs[0][2], imgs[1][2]):
            assert np.array_equal(a, b)

    # ---- silk_find_pitch_lags_FIX
    rec = np.asarray(silk_corpus.corpus(4096, "pitch")["pitch_in"])
    for k in range(0, 4096, 211):
        R = np.frombuffer(np.ascontiguousarray(rec[k]).tobytes(), np.dtype(S.FindPitchLagsIn))[0]

        def mk():
            enc, ctl = _fix_images(d, [(O(n), int(R[f]), i32) for n, f in (
                ("FS_KHZ", "fs_kHz"), ("NB_SUBFR", "nb_subfr"), ("FRAME_LENGTH", "frame_length"), ("LTP_MEM_LENGTH", "ltp_mem_length"),
                ("LA_PITCH", "la_pitch"), ("PITCH_LPC_WIN_LENGTH", "pitch_LPC_win_length"), ("PITCH_EST_LPC_ORDER", "pitchEstimationLPCOrder"),
                ("PITCH_EST_COMPLEXITY", "pitchEstimationComplexity"), ("PITCH_EST_THRESHOLD_Q16", "pitchEstimationThreshold_Q16"),
                ("FIRST_FRAME_AFTER_RESET", "first_frame_after_reset"), ("SPEECH_ACTIVITY_Q8", "speech_activity_Q8"),
                ("INPUT_TILT_Q15", "input_tilt_Q15"), ("PREV_LAG", "prevLag"), ("FIX_LTPCORR_Q15", "LTPCorr_Q15"))] + [
                (IND + O("SIGNAL_TYPE"), int(R["signalType"]), i8), (O("PREV_SIGNAL_TYPE"), int(R["prevSignalType"]), i8)], [])
            xbuf = np.zeros(1024, i16)
            xbuf[:672] = R["x_buf"]
            res = np.full(1024, 0, i16)
            return enc, ctl, (res, xbuf), lambda fn, e, c, x: fn(p(e), p(c), p(x[0]), C.c_void_p(x[1].ctypes.data + 2 * int(R["ltp_mem_length"])), C.c_int(0))
        both(r.silk_find_pitch_lags_FIX, L.opusgpu_silk_find_pitch_lags_FIX, mk)

    # ---- silk_noise_shape_analysis_FIX
    rec = np.asarray(silk_corpus.corpus(4096, "shape")["shape_in"])
    for k in range(0, 4096, 211):
        R = np.frombuffer(np.ascontiguousarray(rec[k]).tobytes(), np.dtype(S.NoiseShapeIn))[0]

        def mk():
            enc, ctl = _fix_images(d, [(O(n), int(R[f]), i32) for n, f in (
                ("FS_KHZ", "fs_kHz"), ("NB_SUBFR", "nb_subfr"), ("SUBFR_LENGTH", "subfr_length"), ("LA_SHAPE", "la_shape"),
                ("SHAPE_WIN_LENGTH", "shapeWinLength"), ("SHAPING_LPC_ORDER", "shapingLPCOrder"), ("WARPING_Q16", "warping_Q16"),
                ("SNR_DB_Q7", "SNR_dB_Q7"), ("USE_CBR", "useCBR"), ("SPEECH_ACTIVITY_Q8", "speech_activity_Q8"), ("FIX_LTPCORR_Q15", "LTPCorr_Q15"))] + [
                (O("INPUT_QUALITY_BANDS_Q15"), R["input_quality_bands_Q15"], i32), (IND + O("SIGNAL_TYPE"), int(R["signalType"]), i8),
                (SHP + O("SHAPE_HARM_BOOST_SMTH_Q16"), int(R["HarmBoost_smth_Q16"]), i32),
                (SHP + O("SHAPE_HARM_SHAPE_GAIN_SMTH_Q16"), int(R["HarmShapeGain_smth_Q16"]), i32),
                (SHP + O("SHAPE_TILT_SMTH_Q16"), int(R["Tilt_smth_Q16"]), i32)],
                [(O("CTRL_PRED_GAIN_Q16"), int(R["predGain_Q16"]), i32), (O("CTRL_PITCHL"), R["pitchL"], i32)])
            x = np.zeros(1024, i16)
            x[:480] = R["x"]
            pr = np.ascontiguousarray(R["pitch_res"]).astype(i16)
            return enc, ctl, (pr, x), lambda fn, e, c, a: fn(p(e), p(c), p(a[0]), C.c_void_p(a[1].ctypes.data + 2 * int(R["la_shape"])), C.c_int(0))
        both(r.silk_noise_shape_analysis_FIX, L.opusgpu_silk_noise_shape_analysis_FIX, mk)

    # ---- silk_process_gains_FIX
    rec = np.asarray(silk_corpus.corpus(4096, "gains")["gains_in"])
    for k in range(0, 4096, 211):
        R = np.frombuffer(np.ascontiguousarray(rec[k]).tobytes(), np.dtype(S.ProcessGainsIn))[0]

        def mk():
            enc, ctl = _fix_images(d, [(O(n), int(R[f]), i32) for n, f in (
                ("NB_SUBFR", "nb_subfr"), ("SUBFR_LENGTH", "subfr_length"), ("SNR_DB_Q7", "SNR_dB_Q7"), ("INPUT_TILT_Q15", "input_tilt_Q15"),
                ("N_STATES_DEL_DEC", "nStatesDelayedDecision"), ("SPEECH_ACTIVITY_Q8", "speech_activity_Q8"))] + [
                (IND + O("SIGNAL_TYPE"), int(R["signalType"]), i8), (IND + O("QUANT_OFFSET_TYPE"), int(R["quantOffsetType"]), i8),
                (SHP + O("SHAPE_LAST_GAIN_INDEX"), int(R["LastGainIndex"]), i8)],
                [(O("CTRL_GAINS_Q16"), R["Gains_Q16"], i32), (O("CTRL_RES_NRG"), R["ResNrg"], i32), (O("CTRL_RES_NRG_Q"), R["ResNrgQ"], i32),
                 (O("CTRL_LTP_RED_COD_GAIN_Q7"), int(R["LTPredCodGain_Q7"]), i32), (O("CTRL_INPUT_QUALITY_Q14"), int(R["input_quality_Q14"]), i32),
                 (O("CTRL_CODING_QUALITY_Q14"), int(R["coding_quality_Q14"]), i32)])
            return enc, ctl, (), lambda fn, e, c, a: fn(p(e), p(c), C.c_int(int(R["condCoding"])))
        both(r.silk_process_gains_FIX, L.opusgpu_silk_process_gains_FIX, mk)

    # ---- silk_prefilter_FIX
    pc = silk_corpus.corpus(4096, "prefilter")
    rec, st0 = np.asarray(pc["prefilter_in"]), np.asarray(pc["prefilter_state_in"])
    for k in range(0, 4096, 211):
        R = np.frombuffer(np.ascontiguousarray(rec[k]).tobytes(), np.dtype(S.PrefilterIn))[0]

        def mk():
            enc, ctl = _fix_images(d, [(O(n), int(R[f]), i32) for n, f in (
                ("NB_SUBFR", "nb_subfr"), ("SUBFR_LENGTH", "subfr_length"), ("WARPING_Q16", "warping_Q16"), ("SHAPING_LPC_ORDER", "shapingLPCOrder"))] + [
                (IND + O("SIGNAL_TYPE"), int(R["signalType"]), i8), (PF, st0[k], np.uint8)],
                [(O("CTRL_PITCHL"), R["pitchL"], i32), (O("CTRL_HARM_SHAPE_GAIN_Q14"), R["HarmShapeGain_Q14"], i32),
                 (O("CTRL_HARM_BOOST_Q14"), R["HarmBoost_Q14"], i32), (O("CTRL_TILT_Q14"), R["Tilt_Q14"], i32),
                 (O("CTRL_GAINS_PRE_Q14"), R["GainsPre_Q14"], i32), (O("CTRL_LF_SHP_Q14"), R["LF_shp_Q14"], i32),
                 (O("CTRL_AR1_Q13"), R["AR1_Q13"], i16), (O("CTRL_CODING_QUALITY_Q14"), int(R["coding_quality_Q14"]), i32)])
            xw = np.zeros(320, i32)
            x = np.ascontiguousarray(R["x"]).astype(i16)
            return enc, ctl, (xw, x), lambda fn, e, c, a: fn(p(e), p(c), p(a[0]), p(a[1]))
        both(r.silk_prefilter_FIX, L.opusgpu_silk_prefilter_FIX, mk)


def test_silk_VAD_hook_with_the_reference_argument_list(L, ref):
    """opusgpu_silk_VAD_GetSA_Q8_c(psEncC, pIn) beside silk_VAD_GetSA_Q8_c of the compiled reference on identical silk_encoder_state
    images (zero except frame_length, fs_kHz and sVAD): every byte of the structure compared afterwards."""
    import silk_corpus
    from concentus_amd import silk as S
    r, _ = ref
    if not silk_corpus.available():
        pytest.skip("capture library did not travel")
    d = header_defines()
    c = silk_corpus.corpus(4096, "vad", complexities=(5,))
    vin, vst = np.asarray(c["vad_in"]), np.asarray(c["vad_state_in"])
    for k in list(range(0, 40)) + list(range(40, 4096, 157)):
        R = np.frombuffer(np.ascontiguousarray(vin[k]).tobytes(), np.dtype(S.VadIn))[0]
        imgs = []
        for fn in (r.silk_VAD_GetSA_Q8_c, L.opusgpu_silk_VAD_GetSA_Q8_c):
            enc = np.zeros(d["OPUSGPU_REF_SIZEOF_SILK_ENCODER_STATE"], np.uint8)
            for name, f in (("FRAME_LENGTH", "frame_length"), ("FS_KHZ", "fs_kHz")):
                o = d["OPUSGPU_REF_OFF_" + name]
                enc[o:o + 4].view(np.int32)[0] = int(R[f])
            enc[d["OPUSGPU_REF_OFF_SVAD"]:][:112] = vst[k]
            pin = np.ascontiguousarray(R["pIn"]).astype(np.int16)
            fn.restype = C.c_int
            assert fn(p(enc), p(pin)) == 0
            imgs.append(enc)
        assert L.opusgpu_get_last_error() == 0
        assert np.array_equal(imgs[0], imgs[1]), (k, np.nonzero(imgs[0] != imgs[1])[0][:8])


def test_quant_all_bands_hook_on_the_reference_encoders_own_calls(L, ref):
    """opusgpu_quant_all_bands with the tree's 21-argument list and its ec_ctx (EC_DIFF included) against quant_all_bands of the
    compiled reference ON THE CALLS THE REFERENCE ENCODER ITSELF MAKES: oracle/_ref/libopus_ref_celtcap.so (--wrap=quant_all_bands,
    oracle/ref_celt_capture.c) records the arguments before each call and the range coder after it while opus_encode() runs over
    noise, music and the reference's own test signal at several rates; compared: every ec_ctx field and every byte of the
    coder's buffer."""
    import encode_cases as ec
    cap_path = os.path.join(ROOT, "oracle", "_ref", "libopus_ref_celtcap.so")
    if not os.path.exists(cap_path):
        pytest.skip("oracle/_ref/libopus_ref_celtcap.so did not travel")
    cap = C.CDLL(cap_path)
    cap.opus_encoder_create.restype = C.c_void_p
    cap.opus_encoder_ctl.argtypes = [C.c_void_p, C.c_int, C.c_int]
    gm = ec.golden_module()
    cap.refcap_start_qab(4096)
    for kind, br, vbr, cx, seed in (("noise", 96000, 1, 10, 71), ("music", 64000, 1, 10, 72), ("gmusic", 128000, 0, 5, 13371337),
                                    ("music", 40000, 1, 8, 73)):
        pcm = gm.synth_pcm(kind, 12, seed)
        err = C.c_int()
        enc = C.c_void_p(cap.opus_encoder_create(48000, 2, 2051, C.byref(err)))
        for req, v in ((4002, br), (4006, vbr), (4020, 0), (4010, cx), (4036, 16)):
            cap.opus_encoder_ctl(enc, req, v)
        out = (C.c_ubyte * 1500)()
        for f in range(pcm.shape[0]):
            fr = np.ascontiguousarray(pcm[f])
            assert cap.opus_encode(enc, p(fr), 960, out, 1500) > 0
    n = cap.refcap_count_qab()
    assert n == 48
    sz = cap.refcap_sizeof_qab()
    raw = np.zeros((n, sz), np.uint8)
    cap.refcap_get_qab(p(raw))
    mode = reflib.lib().opus_custom_mode_create(48000, 960, C.byref(C.c_int()))
    o = 0
    fields = {}
    for name, nbytes in (("X", 1920), ("Y", 1920), ("bandE", 168), ("pulses", 84), ("tf_res", 84), ("ints", 40), ("tb", 8), ("seed", 4),
                         ("ec_in", 44), ("ec_out", 44), ("buf_in", 1280), ("buf_out", 1280)):
        fields[name] = (o, nbytes)
        o += nbytes
    assert o <= sz
    get = lambda r, k, dt: np.ascontiguousarray(raw[r, fields[k][0]:fields[k][0] + fields[k][1]]).view(dt).copy()
    seen_short = set()
    for r in range(n):
        X, Y = get(r, "X", np.int16), get(r, "Y", np.int16)
        bandE, pulses, tf_res = get(r, "bandE", np.int32), get(r, "pulses", np.int32), get(r, "tf_res", np.int32)
        encode, start, end, _st, shortBlocks, spread, dual, intensity, LM, coded = [int(v) for v in get(r, "ints", np.int32)]
        total_bits, balance = [int(v) for v in get(r, "tb", np.int32)]
        seed = C.c_uint32(int(get(r, "seed", np.uint32)[0]))
        ein, eout = get(r, "ec_in", np.int32), get(r, "ec_out", np.int32)
        buf = get(r, "buf_in", np.uint8)
        want_buf = get(r, "buf_out", np.uint8)
        seen_short.add(shortBlocks)
        e = reflib.EcCtx()
        e.buf = buf.ctypes.data_as(C.POINTER(C.c_ubyte))
        (e.storage, e.end_offs, e.end_window, e.nend_bits, e.nbits_total, e.offs, e.rng, e.val, e.ext, e.rem, e.error) = \
            [int(v) & 0xffffffff if i in (0, 1, 2, 5, 6, 7, 8) else int(v) for i, v in enumerate(ein)]
        e.EC_DIFF = 0
        cm = np.zeros(42, np.uint8)
        L.opusgpu_quant_all_bands(encode, mode, start, end, p(X), p(Y), p(cm), p(bandE), p(pulses), shortBlocks, spread, dual, intensity,
                                  p(tf_res), total_bits, balance, C.byref(e), LM, coded, C.byref(seed), 0)
        assert L.opusgpu_get_last_error() == 0, r
        got = np.array([e.storage, e.end_offs, e.end_window, e.nend_bits, e.nbits_total, e.offs, e.rng, e.val, e.ext, e.rem, e.error],
                       dtype=np.int64) & 0xffffffff
        assert np.array_equal(got, eout.astype(np.int64) & 0xffffffff), (r, got, eout)
        st = int(e.storage)
        assert np.array_equal(buf[:st], want_buf[:st]), (r, np.nonzero(buf[:st] != want_buf[:st])[0][:8])
    assert seen_short == {0, 8}


def test_quant_all_bands_hook_decode_side_on_the_reference_decoders_own_calls(L, ref):
    """opusgpu_quant_all_bands(encode = 0, ...) against quant_all_bands of the compiled reference ON THE CALLS THE REFERENCE DECODER
    ITSELF MAKES (celt_decoder.c:977): oracle/_ref/libopus_ref_celtcap.so records the arguments and the range decoder before each
    call and, after it, the decoded normalised bands, the collapse masks, the LCG seed and the range decoder while opus_decode()
    runs over packets of several rates (32 kb/s: folding, noise fill, intensity stereo; 96 / 128 kb/s; transient frames).
    Compared: X / Y over the coded bins, all 42 collapse masks, *seed, every ec_ctx field."""
    import encode_cases as ec
    cap_path = os.path.join(ROOT, "oracle", "_ref", "libopus_ref_celtcap.so")
    if not os.path.exists(cap_path):
        pytest.skip("oracle/_ref/libopus_ref_celtcap.so did not travel")
    cap = C.CDLL(cap_path)
    if not hasattr(cap, "refcap_start_qab_dec"):
        pytest.skip("capture library predates the decoder-side capture")
    cap.opus_encoder_create.restype = C.c_void_p
    cap.opus_decoder_create.restype = C.c_void_p
    cap.opus_encoder_ctl.argtypes = [C.c_void_p, C.c_int, C.c_int]
    cap.opus_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int, C.c_int]
    gm = ec.golden_module()
    cap.refcap_start_qab_dec(4096)
    nframes = 0
    for kind, br, vbr, cx, seed in (("noise", 96000, 1, 10, 81), ("music", 32000, 1, 10, 82), ("gmusic", 128000, 0, 5, 13371337),
                                    ("music", 40000, 1, 8, 83), ("noise", 36000, 0, 10, 84)):
        pcm = gm.synth_pcm(kind, 12, seed)
        err = C.c_int()
        enc = C.c_void_p(cap.opus_encoder_create(48000, 2, 2051, C.byref(err)))
        dec = C.c_void_p(cap.opus_decoder_create(48000, 2, C.byref(err)))
        for req, v in ((4002, br), (4006, vbr), (4020, 0), (4010, cx), (4036, 16)):
            cap.opus_encoder_ctl(enc, req, v)
        out = (C.c_ubyte * 1500)()
        dpcm = np.zeros((960, 2), np.int16)
        for f in range(pcm.shape[0]):
            fr = np.ascontiguousarray(pcm[f])
            nb = cap.opus_encode(enc, p(fr), 960, out, 1500)
            assert nb > 0
            assert cap.opus_decode(dec, out, nb, p(dpcm), 960, 0) == 960
            nframes += 1
    n = cap.refcap_count_qab_dec()
    assert n == nframes == 60
    sz = cap.refcap_sizeof_qab_dec()
    raw = np.zeros((n, sz), np.uint8)
    cap.refcap_get_qab_dec(p(raw))
    mode = reflib.lib().opus_custom_mode_create(48000, 960, C.byref(C.c_int()))
    o = 0
    fields = {}
    for name, nbytes in (("pulses", 84), ("tf_res", 84), ("ints", 24), ("tb", 8), ("seeds", 8), ("ec_in", 44), ("ec_out", 44),
                         ("buf", 1280), ("X_out", 1920), ("Y_out", 1920), ("cm_out", 42)):
        fields[name] = (o, nbytes)
        o += nbytes
    assert o <= sz
    get = lambda r, k, dt: np.ascontiguousarray(raw[r, fields[k][0]:fields[k][0] + fields[k][1]]).view(dt).copy()
    seen_short, seen_dual, folded = set(), set(), 0
    for r in range(n):
        pulses, tf_res = get(r, "pulses", np.int32), get(r, "tf_res", np.int32)
        shortBlocks, spread, dual, intensity, LM, coded = [int(v) for v in get(r, "ints", np.int32)]
        total_bits, balance = [int(v) for v in get(r, "tb", np.int32)]
        seed_in, seed_out = [int(v) for v in get(r, "seeds", np.uint32)]
        ein, eout = get(r, "ec_in", np.int32), get(r, "ec_out", np.int32)
        buf = get(r, "buf", np.uint8)
        seen_short.add(shortBlocks)
        seen_dual.add(dual)
        folded += int((pulses[:coded] <= 0).any())
        e = reflib.EcCtx()
        e.buf = buf.ctypes.data_as(C.POINTER(C.c_ubyte))
        (e.storage, e.end_offs, e.end_window, e.nend_bits, e.nbits_total, e.offs, e.rng, e.val, e.ext, e.rem, e.error) = \
            [int(v) & 0xffffffff if i in (0, 1, 2, 5, 6, 7, 8) else int(v) for i, v in enumerate(ein)]
        e.EC_DIFF = 0
        X, Y = np.full(960, 0x5555, np.int16), np.full(960, 0x5555, np.int16)
        cm = np.full(42, 0xEE, np.uint8)
        seed = C.c_uint32(seed_in)
        L.opusgpu_quant_all_bands(0, mode, 0, 21, p(X), p(Y), p(cm), None, p(pulses), shortBlocks, spread, dual, intensity,
                                  p(tf_res), total_bits, balance, C.byref(e), LM, coded, C.byref(seed), 0)
        assert L.opusgpu_get_last_error() == 0, r
        got = np.array([e.storage, e.end_offs, e.end_window, e.nend_bits, e.nbits_total, e.offs, e.rng, e.val, e.ext, e.rem, e.error],
                       dtype=np.int64) & 0xffffffff
        assert np.array_equal(got, eout.astype(np.int64) & 0xffffffff), (r, got, eout)
        assert seed.value == seed_out, (r, seed.value, seed_out)
        assert np.array_equal(cm, get(r, "cm_out", np.uint8)), (r, cm, get(r, "cm_out", np.uint8))
        wx, wy = get(r, "X_out", np.int16), get(r, "Y_out", np.int16)
        assert np.array_equal(X[:800], wx[:800]), (r, np.nonzero(X[:800] != wx[:800])[0][:8])
        assert np.array_equal(Y[:800], wy[:800]), (r, np.nonzero(Y[:800] != wy[:800])[0][:8])
    assert seen_short == {0, 8} and folded > 0, (seen_short, seen_dual, folded)


GPUFRAME = os.path.join(ROOT, "oracle", "_ref", "libopus_ref_gpuframe.so")


def _encode_all(lib, pcm, fs, frame, ctls, max_bytes=1500):
    lib.opus_encoder_create.restype = C.c_void_p
    lib.opus_encoder_ctl.argtypes = [C.c_void_p, C.c_int, C.c_int]
    err = C.c_int()
    enc = C.c_void_p(lib.opus_encoder_create(fs, 1, 2048, C.byref(err)))          # OPUS_APPLICATION_VOIP
    assert enc and err.value == 0
    for req, v in ctls:
        assert lib.opus_encoder_ctl(enc, req, v) == 0
    out = (C.c_ubyte * 1500)()
    rng = C.c_uint32()
    lib.opus_encoder_ctl.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    packets = []
    for f in range(len(pcm) // frame):
        fr = np.ascontiguousarray(pcm[f * frame:(f + 1) * frame])
        n = lib.opus_encode(enc, p(fr), frame, out, max_bytes)
        assert n > 0, (f, n)
        assert lib.opus_encoder_ctl(enc, 4031, C.byref(rng)) == 0                  # OPUS_GET_FINAL_RANGE
        packets.append((bytes(out[:n]), rng.value))
    return packets


@pytest.mark.parametrize("name,fs,frame,ctls,max_bytes,mask", [
    ("wb 32k vbr cx10", 16000, 320, ((4002, 32000), (4006, 1), (4020, 0), (4010, 10)), 1500, 3),
    ("wb 24k cbr cx5", 16000, 320, ((4002, 24000), (4006, 0), (4010, 5)), 1500, 3),
    ("nb 12k cbr cx3", 8000, 160, ((4002, 12000), (4006, 0), (4010, 3)), 1500, 1),
    ("wb 10 ms vbr cx8 squeezed", 16000, 160, ((4002, 40000), (4006, 1), (4020, 0), (4010, 8)), 40, 3),
    ("wb 40 ms vbr cx7", 16000, 640, ((4002, 28000), (4006, 1), (4020, 0), (4010, 7)), 1500, 3),
    # 12 kHz (mediumband) lies outside the hook's operating region: it must say OPUSGPU_UNIMPLEMENTED with the encoder state
    # untouched, the wrap shim then runs the reference's own function on the frame -- same packets, no failures
    ("mb 12 kHz vbr cx6: every frame declined", 12000, 240, ((4002, 20000), (4006, 1), (4020, 0), (4010, 6)), 1500, 1),
])
def test_reference_encoder_with_its_silk_frame_function_replaced_emits_the_same_packets(L, name, fs, frame, ctls, max_bytes, mask):
    """The drop-in claim at packet level: opus_encode() of the UNMODIFIED reference, linked with --wrap so that every call of
    silk_encode_frame_FIX (and of silk_VAD_GetSA_Q8_c, mask 3) lands in libopusgpu.so's hooks of the same argument lists
    (oracle/ref_gpuframe_wrap.c), against opus_encode() of the plain reference on the same speech-like PCM: every packet byte and
    the final range, VBR / CBR (the bitrate loop), wideband / narrowband, 10 / 20 / 40 ms packets (conditional coding)."""
    import silk_corpus
    import concentus_amd
    if not (reflib.available() and os.path.exists(GPUFRAME)):
        pytest.skip("oracle/_ref did not travel")
    nframes = 60 * 320 // frame if fs == 16000 else 60
    if fs == 12000:
        v = silk_corpus.synth_voice(nframes * 320 + 16000, 424242)[16000:].astype(np.float64)                   # 16 kHz -> 12 kHz, linear
        t = np.arange(nframes * frame) * (16000.0 / 12000.0)
        pcm = np.interp(t, np.arange(len(v)), v).astype(np.int16)
    else:
        pcm = silk_corpus.synth_voice(nframes * frame * (16000 // fs) + 16000, 424242)[16000:][::16000 // fs]  # skip the leading pause
    common = ((4012, 0), (4016, 0), (4014, 0), (4036, 16))
    want = _encode_all(C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libopus_ref.so")), pcm, fs, frame, ctls + common, max_bytes)
    hooked = C.CDLL(GPUFRAME)
    assert hooked.refgpu_load(concentus_amd.lib.LIB_PATH.encode(), mask) == 0
    got = _encode_all(hooked, pcm, fs, frame, ctls + common, max_bytes)
    assert hooked.refgpu_failures() == 0, (name, hooked.refgpu_failures(), hooked.refgpu_first_error())
    assert hooked.refgpu_calls(0) >= nframes * (frame // (fs // 50) if frame > fs // 50 else 1), "every SILK frame went through the hook"
    hooked.refgpu_fallbacks.restype = C.c_int
    if fs == 12000:
        assert hooked.refgpu_fallbacks() == hooked.refgpu_calls(0), "12 kHz frames are declined before any state is touched"
    else:
        assert hooked.refgpu_fallbacks() == 0, "inside the operating region nothing falls back to the reference"
    if mask & 2:
        assert hooked.refgpu_calls(1) >= nframes
    diff = [k for k in range(len(want)) if want[k] != got[k]]
    assert not diff, (name, len(diff), diff[:8])
    assert len(set(w[0] for w in want)) > len(want) // 2 and np.mean([len(w[0]) for w in want]) > 20, "real payloads"

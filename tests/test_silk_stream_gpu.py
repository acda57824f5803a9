"""GPU tier: STREAMS MODE of the SILK frame chain (from PCM across frames). S streams x T consecutive frames run
through the device chain with the inter-frame state carried ON THE DEVICE (opusgpu_silk_stream_carry_in / _out around
opusgpu_silk_encode_frames(_cbr)_batch); the reference's capture of the SAME run supplies (a) the stream state at t = 0 only and (b)
per frame what is computed outside silk_encode_frame_FIX -- the frame's samples, the VAD results, SNR_dB_Q7, maxBits / condCoding and
the packet's fresh range coder. Every carried field of every record is ZEROED on the host before upload (CARRIED_FIELDS), so a frame
can only be right if the device carried the state. Parity per frame: the range coder after the frame (every field, every payload
byte), silk_nsq_state, the prefilter state, pulses, Seed, LastGainIndex, number of passes -- against what silk_encode_frame_FIX left
behind in the capture."""
import os
import sys

import numpy as np
import pytest

import silk_corpus

pytestmark = pytest.mark.gpu


def stream_state_from_capture(rec, rows, S, dd):
    """opusgpu_silk_stream records for frame `rows` of the capture: what the frame BEFORE it left behind, read off this frame's inputs."""
    n = len(rows)
    st = np.zeros(n, np.dtype(S.SilkStream))
    P = np.ascontiguousarray(rec["c_pitch_in"][rows]).view(np.dtype(S.FindPitchLagsIn))[:, 0]
    SI = np.ascontiguousarray(rec["c_shape_in"][rows]).view(np.dtype(S.NoiseShapeIn))[:, 0]
    FI = np.ascontiguousarray(rec["c_fpc_in"][rows]).view(np.dtype(S.FindPredCoefsIn))[:, 0]
    GI = np.ascontiguousarray(rec["c_gains_in"][rows]).view(np.dtype(S.ProcessGainsIn))[:, 0]
    BI = np.ascontiguousarray(rec["c_bits_in"][rows]).view(np.dtype(S.SilkBitsIn))[:, 0]
    Q = np.ascontiguousarray(rec["c_q_in"][rows]).view(np.dtype(S.NsqDdIn if dd else S.NsqIn))[:, 0]
    Q = Q["base"] if dd else Q
    fs, nb = int(P["fs_kHz"][0]), int(P["nb_subfr"][0])
    ltp, la_s = 20 * fs, 5 * fs
    st["x_buf"][:, :ltp + la_s] = P["x_buf"][:, :ltp + la_s]
    st["prev_NLSFq_Q15"] = FI["prev_NLSFq_Q15"]
    for f in ("prevLag", "prevSignalType", "first_frame_after_reset", "LTPCorr_Q15"):
        st[f] = P[f]
    st["sum_log_gain_Q7"] = FI["sum_log_gain_Q7"]
    st["LastGainIndex"] = GI["LastGainIndex"]
    for f in ("HarmBoost_smth_Q16", "HarmShapeGain_smth_Q16", "Tilt_smth_Q16"):
        st[f] = SI[f]
    st["ec_prevSignalType"], st["ec_prevLagIndex"] = BI["ec_prevSignalType"], BI["ec_prevLagIndex"]
    st["frameCounter"] = Q["Seed"]                          # only frameCounter & 3 is ever read
    return st


def zero_fields(host, table):
    for name, (cls, fields) in table.items():
        for f in fields:
            d = getattr(cls, f)
            host[name][:, d.offset:d.offset + d.size] = 0


@pytest.mark.parametrize("kind,variant,T", [("chain_dd", "wb20cbr", 24), ("chain_nsq", "nb20cbr", 16), ("chain_dd", "wb20", 24)])
def test_streams_of_consecutive_frames_match_the_reference_with_state_carried_on_the_device(kind, variant, T):
    import torch
    import concentus_amd as ca
    from concentus_amd import silk as S
    from concentus_amd.silk_chain import SilkAnalysisChain, CHAIN_FED_FIELDS, CARRIED_FIELDS
    from test_silk_rate_cpu import fresh_ctl
    if not silk_corpus.available():
        pytest.skip("capture library did not travel")
    streams_n = 96
    n = streams_n * T
    rec = {k: np.array(v) for k, v in silk_corpus.corpus(n, kind, variant=variant, seg_frames=T).items()}
    dd = kind == "chain_dd"
    fs = 8 if variant.startswith("nb") else 16
    chain = SilkAnalysisChain(fs, 4)
    fl, ltp, la_s = chain.frame_length, 20 * fs, 5 * fs
    names = {"pitch_in": "c_pitch_in", "shape_in": "c_shape_in", "fpc_in": "c_fpc_in", "gains_in": "c_gains_in", "prefilter_in": "c_prefilter_in",
             "q_in": "c_q_in", "bits_in": "c_bits_in"}
    rows0 = np.arange(streams_n) * T
    st = torch.from_numpy(stream_state_from_capture(rec, rows0, S, dd).view(np.uint8).reshape(streams_n, -1).copy()).cuda()
    pf = torch.from_numpy(rec["c_prefilter_state_in"][rows0].copy()).cuda()           # the two big states: captured at t = 0, then the device's own
    nsq = torch.from_numpy(rec["c_q_state_in"][rows0].copy()).cuda()
    passes_hist = np.zeros(9, np.int64)
    voiced = 0
    for t in range(T):
        rows = rows0 + t
        host = {k: rec[v][rows].copy() for k, v in names.items()}
        # the frame's samples = the new part of x_buf (what encode_frame_FIX.c:145 copies in), read off the shaping record
        SI = host["shape_in"].view(np.dtype(S.NoiseShapeIn))[:, 0]
        frame_input = np.zeros((streams_n, 320), np.int16)
        frame_input[:, :fl] = SI["x"][:, 2 * la_s:2 * la_s + fl]
        zero_fields(host, CHAIN_FED_FIELDS)
        zero_fields(host, CARRIED_FIELDS)
        sub = {k: rec[k][rows] for k in ("c_frame_args", "c_gains_out", "c_gains_in")}
        ctl = fresh_ctl(sub, streams_n)
        for f in ("GainsUnq_Q16", "Gains_Q16", "lastGainIndexPrev", "LastGainIndex", "Lambda_Q10", "GainsIndices"):
            ctl[f] = 0
        dev = {k: torch.from_numpy(v).cuda() for k, v in host.items()}
        rate_ctl = torch.from_numpy(ctl.view(np.uint8).reshape(streams_n, -1).copy()).cuda()
        ec = torch.from_numpy(rec["c_ec_in"][rows].copy()).cuda()                      # the packet's coder as silk_Encode hands it over
        out = chain.run(dev["pitch_in"], dev["shape_in"], dev["fpc_in"], dev["gains_in"], dev["prefilter_in"], pf, dev["q_in"], nsq, dd,
                        bits_in=dev["bits_in"], ec_state=ec, rate_ctl=rate_ctl, streams=st, frame_input=torch.from_numpy(frame_input).cuda())
        torch.cuda.synchronize()
        got_ctl = rate_ctl.cpu().numpy().view(np.dtype(S.RateCtl))[:, 0]
        args, misc = rec["c_frame_args"][rows].view(np.int32), rec["c_frame_misc"][rows]
        assert (got_ctl["done"] == 1).all() and (got_ctl["status"] == 0).all(), t
        got = ec.cpu().numpy()
        bad = np.nonzero((got != rec["c_frame_ec"][rows]).any(1))[0]
        assert bad.size == 0, ("range coder after frame", t, bad.size, bad[:6])
        assert np.array_equal(nsq.cpu().numpy(), rec["c_frame_nsq"][rows]), ("silk_nsq_state", t)
        assert np.array_equal(pf.cpu().numpy(), rec["c_prefilter_state_out"][rows]), ("prefilter state", t)
        assert np.array_equal(out["pulses"].cpu().numpy().view(np.uint8), misc[:, :320]), ("pulses", t)
        assert np.array_equal(got_ctl["passes"], args[:, 3]) and np.array_equal(got_ctl["LastGainIndex"], misc[:, 324:328].copy().view(np.int32)[:, 0])
        if dd:
            assert np.array_equal(out["Seed"].cpu().numpy(), misc[:, 328:332].copy().view(np.int32)[:, 0]), ("Seed", t)
        passes_hist += np.bincount(args[:, 3], minlength=9)[:9]
        voiced += int((out["pitch_out"].cpu().numpy().view(np.dtype(S.FindPitchLagsOut))[:, 0]["signalType"] == 2).sum())
        if t + 1 < T:
            # the stream records the device now holds = what the capture says the next frame inherits
            want = stream_state_from_capture(rec, rows + 1, S, dd)
            have = st.cpu().numpy().view(np.dtype(S.SilkStream))[:, 0]
            for f in want.dtype.names:
                if f in ("reserved",):
                    continue
                a, b = (have[f] & 3, want[f] & 3) if f == "frameCounter" else (have[f], want[f])
                if f == "x_buf":
                    a, b = a[:, :ltp + la_s], b[:, :ltp + la_s]
                if f == "prev_NLSFq_Q15":
                    D = 10 if fs == 8 else 16
                    a, b = a[:, :D], b[:, :D]
                assert np.array_equal(a, b), ("stream record", f, t)
    assert voiced > streams_n and (passes_hist[2:].sum() > n // 4 if variant.endswith("cbr") else passes_hist[1] > 0), (voiced, passes_hist)

"""GPU tier: the SILK analysis chain END TO END on the device: from the pitch-analysis buffer of a frame to the
quantiser's pulses through the seven batched kernels, the records between them filled on the device by
concentus_amd.silk_chain.SilkAnalysisChain (byte moves along the edges tests/test_silk_chain_cpu.py pins). Input: the records of
ONE run of the unmodified reference encoder captured with frame numbers; every field the chain is supposed to fill is ZEROED
before the run. Compared with what the reference computed for the same frames: every stage's output record, the prefilter
state, the pulses, Seed and every byte of silk_nsq_state."""
import numpy as np
import pytest

import silk_corpus
from test_silk_chain_cpu import _capture, _first_of_frame

pytestmark = pytest.mark.gpu


def _bytes(a):
    return np.ascontiguousarray(a).view(np.uint8).reshape(a.shape[0], -1)


@pytest.mark.parametrize("complexity,variant", [(3, "wb20"), (5, "wb20"), (10, "wb20"), (8, "nb20"), (6, "wb10")])
def test_chain_from_pitch_buffer_to_pulses(complexity, variant):
    import torch
    import concentus_amd as ca
    from concentus_amd import silk as S
    from concentus_amd.silk_chain import SilkAnalysisChain, CHAIN_FED_FIELDS
    if not silk_corpus.available():
        pytest.skip("capture library did not travel")
    nframes = 700
    r = _capture(complexity, nframes, 9000 + complexity, variant)
    (pin, pout, pf), (sin_, sout, sf), (fin, fout, ff), (gin, gout, gf), (xin, xst0, xst1, xout, xf), (_, qf) = (
        r["pitch"], r["shape"], r["fpc"], r["gains"], r["prefilter"], r["q"])
    qin, qst0, qst1, qout, del_dec = r["q_full"]
    idx = {k: [] for k in "psfgxq"}
    for frame in range(9, nframes + 1):
        ks = [_first_of_frame(a, frame) for a in (pf, sf, ff, gf, xf, qf)]
        assert None not in ks
        for key, k in zip("psfgxq", ks):
            idx[key].append(k)
    sel = {k: np.array(v) for k, v in idx.items()}
    rec = {"pitch_in": _bytes(pin[sel["p"]]), "shape_in": _bytes(sin_[sel["s"]]), "fpc_in": _bytes(fin[sel["f"]]), "gains_in": _bytes(gin[sel["g"]]),
           "prefilter_in": _bytes(xin[sel["x"]]), "q_in": _bytes(qin[sel["q"]])}
    # zero every field the chain fills
    for name, (cls, fields) in CHAIN_FED_FIELDS.items():
        if name not in rec:                                   # (the entropy-coding stage has its own test below)
            continue
        for f in fields:
            d = getattr(cls, f)
            rec[name][:, d.offset:d.offset + d.size] = 0
    dev = {k: torch.from_numpy(v.copy()).cuda() for k, v in rec.items()}
    pf_state = torch.from_numpy(_bytes(xst0[sel["x"]]).copy()).cuda()
    nsq_state = torch.from_numpy(_bytes(qst0[sel["q"]]).copy()).cuda()
    out = SilkAnalysisChain(8 if variant == "nb20" else 16, 2 if variant == "wb10" else 4).run(dev["pitch_in"], dev["shape_in"], dev["fpc_in"], dev["gains_in"], dev["prefilter_in"], pf_state,
                                       dev["q_in"], nsq_state, del_dec)
    torch.cuda.synchronize()
    assert ca.silk.bad_records() == 0
    n = len(sel["p"])
    for key, want, nb in (("pitch_out", pout[sel["p"]], 1380), ("shape_out", sout[sel["s"]], 380), ("fpc_out", fout[sel["f"]], 204),
                          ("gains_out", gout[sel["g"]], 52), ("prefilter_out", xout[sel["x"]], 1280)):
        got = out[key].cpu().numpy()
        bad = np.nonzero((got[:, :nb] != _bytes(want)[:, :nb]).any(1))[0]
        assert bad.size == 0, (key, bad.size, bad[:6])
    assert np.array_equal(pf_state.cpu().numpy(), _bytes(xst1[sel["x"]]))
    want_q = qout[sel["q"]]
    assert np.array_equal(out["pulses"].cpu().numpy().view(np.uint8), want_q[:, :320]), "pulses"
    if del_dec:
        assert np.array_equal(out["Seed"].cpu().numpy(), want_q[:, 320:324].copy().view(np.int32)[:, 0]), "Seed"
    assert np.array_equal(nsq_state.cpu().numpy(), _bytes(qst1[sel["q"]])), "silk_nsq_state"
    voiced = (pout[sel["p"]]["signalType"] == 2).sum()
    assert n > 600 and 50 < voiced < n


@pytest.mark.parametrize("kind,variant", [("chain_dd", "wb20"), ("chain_nsq", "wb20"), ("chain_dd", "wb40"), ("chain_dd", "nb20"), ("chain_dd", "wb10")])
def test_chain_from_pitch_buffer_to_range_coder_bytes(kind, variant):
    """The same chain one stage further: silk_encode_indices + silk_encode_pulses on the frame's range coder (tests/silk_corpus.py
    aligned corpus, 4 096 frames): from the pitch buffer and the coder as it was when the reference reached silk_encode_indices, to
    the coder after silk_encode_pulses -- every ec_ctx field and every byte written -- with all chain-fed record fields zeroed."""
    import torch
    import concentus_amd as ca
    from concentus_amd.silk_chain import SilkAnalysisChain, CHAIN_FED_FIELDS
    if not silk_corpus.available():
        pytest.skip("capture library did not travel")
    rec = {k: np.array(v) for k, v in silk_corpus.corpus(4096, kind, variant=variant).items()}
    names = {"pitch_in": "c_pitch_in", "shape_in": "c_shape_in", "fpc_in": "c_fpc_in", "gains_in": "c_gains_in", "prefilter_in": "c_prefilter_in",
             "q_in": "c_q_in", "bits_in": "c_bits_in"}
    host = {k: rec[v].copy() for k, v in names.items()}
    for name, (cls, fields) in CHAIN_FED_FIELDS.items():
        for f in fields:
            d = getattr(cls, f)
            host[name][:, d.offset:d.offset + d.size] = 0
    dev = {k: torch.from_numpy(v).cuda() for k, v in host.items()}
    pf, nsq, ec = (torch.from_numpy(rec[k]).cuda() for k in ("c_prefilter_state_in", "c_q_state_in", "c_ec_in"))
    ca.silk.bad_records()
    out = SilkAnalysisChain(8 if variant == "nb20" else 16, 2 if variant == "wb10" else 4).run(
        dev["pitch_in"], dev["shape_in"], dev["fpc_in"], dev["gains_in"], dev["prefilter_in"], pf, dev["q_in"], nsq, kind == "chain_dd",
        bits_in=dev["bits_in"], ec_state=ec)
    torch.cuda.synchronize()
    assert ca.silk.bad_records() == 0
    assert np.array_equal(out["pulses"].cpu().numpy().view(np.uint8), rec["c_q_out"][:, :320])
    assert np.array_equal(nsq.cpu().numpy(), rec["c_q_state_out"]) and np.array_equal(pf.cpu().numpy(), rec["c_prefilter_state_out"])
    got = ec.cpu().numpy()
    bad = np.nonzero((got != rec["c_ec_out"]).any(1))[0]
    assert bad.size == 0, (bad.size, bad[:6], np.nonzero(got[bad[0]] != rec["c_ec_out"][bad[0]])[0][:12])
    assert np.array_equal(out["bits_out"].cpu().numpy()[:, :8], rec["c_bits_out"][:, :8])
    written = rec["c_ec_out"][:, 20:24].copy().view(np.uint32)[:, 0] - rec["c_ec_in"][:, 20:24].copy().view(np.uint32)[:, 0]
    assert written.mean() > 20, "the frames' payload bytes"


@pytest.mark.parametrize("kind,variant", [("chain_dd", "wb20cbr"), ("chain_nsq", "nb20cbr"), ("chain_dd", "wb20lo")])
def test_chain_with_bitrate_loop_matches_silk_encode_frame(kind, variant):
    """silk_encode_frame_FIX whole, after the VAD: the chain with the bitrate loop (opusgpu_silk_rate_control_batch after every pass,
    re-quantising and re-coding the frames over / under budget from their entry states) against what the reference's function left
    behind -- the range coder (fields + bytes), silk_nsq_state, the pulses, Seed, LastGainIndex, the gain indices and the number of
    passes. Constant-bitrate encoders iterate on nearly every frame (1-7 passes); "wb20lo" is VBR squeezed by max_data_bytes."""
    import torch
    import concentus_amd as ca
    from concentus_amd import silk as S
    from concentus_amd.silk_chain import SilkAnalysisChain, CHAIN_FED_FIELDS
    from test_silk_rate_cpu import fresh_ctl
    if not silk_corpus.available():
        pytest.skip("capture library did not travel")
    n = 4096
    rec = {k: np.array(v) for k, v in silk_corpus.corpus(n, kind, variant=variant).items()}
    names = {"pitch_in": "c_pitch_in", "shape_in": "c_shape_in", "fpc_in": "c_fpc_in", "gains_in": "c_gains_in", "prefilter_in": "c_prefilter_in",
             "q_in": "c_q_in", "bits_in": "c_bits_in"}
    host = {k: rec[v].copy() for k, v in names.items()}
    for name, (cls, fields) in CHAIN_FED_FIELDS.items():
        for f in fields:
            d = getattr(cls, f)
            host[name][:, d.offset:d.offset + d.size] = 0
    ctl = fresh_ctl(rec, n)
    for f in ("GainsUnq_Q16", "Gains_Q16", "lastGainIndexPrev", "LastGainIndex", "Lambda_Q10", "GainsIndices"):      # the chain fills them
        ctl[f] = 0
    dev = {k: torch.from_numpy(v).cuda() for k, v in host.items()}
    rate_ctl = torch.from_numpy(ctl.view(np.uint8).reshape(n, -1).copy()).cuda()
    pf, nsq, ec = (torch.from_numpy(rec[k]).cuda() for k in ("c_prefilter_state_in", "c_q_state_in", "c_ec_in"))
    ca.silk.bad_records()
    out = SilkAnalysisChain(8 if variant.startswith("nb") else 16, 4).run(
        dev["pitch_in"], dev["shape_in"], dev["fpc_in"], dev["gains_in"], dev["prefilter_in"], pf, dev["q_in"], nsq, kind == "chain_dd",
        bits_in=dev["bits_in"], ec_state=ec, rate_ctl=rate_ctl)
    torch.cuda.synchronize()
    assert ca.silk.bad_records() == 0
    got_ctl = rate_ctl.cpu().numpy().view(np.dtype(S.RateCtl))[:, 0]
    args, misc = rec["c_frame_args"].view(np.int32), rec["c_frame_misc"]
    assert (got_ctl["done"] == 1).all() and (got_ctl["status"] == 0).all()
    assert np.array_equal(got_ctl["passes"], args[:, 3]), "quantiser passes per frame"
    got = ec.cpu().numpy()
    bad = np.nonzero((got != rec["c_frame_ec"]).any(1))[0]
    assert bad.size == 0, (bad.size, bad[:6], args[bad[:6], 3], np.nonzero(got[bad[0]] != rec["c_frame_ec"][bad[0]])[0][:12])
    assert np.array_equal(nsq.cpu().numpy(), rec["c_frame_nsq"]), "silk_nsq_state"
    assert np.array_equal(out["pulses"].cpu().numpy().view(np.uint8), misc[:, :320]), "pulses"
    assert np.array_equal(got_ctl["LastGainIndex"], misc[:, 324:328].copy().view(np.int32)[:, 0])
    assert np.array_equal(got_ctl["GainsIndices"].view(np.uint8), misc[:, 320:324])
    if kind == "chain_dd":
        assert np.array_equal(out["Seed"].cpu().numpy(), misc[:, 328:332].copy().view(np.int32)[:, 0]), "Seed"
    assert np.array_equal(pf.cpu().numpy(), rec["c_prefilter_state_out"])
    hist = np.bincount(args[:, 3], minlength=8)
    assert hist[2:].sum() > (n // 2 if variant.endswith("cbr") else 40), hist
    if variant.endswith("cbr"):
        assert (got_ctl["found_lower"] & got_ctl["found_upper"]).sum() > 100, "frames that bracketed the budget from both sides"

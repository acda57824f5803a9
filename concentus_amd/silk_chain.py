"""The SILK analysis chain of one frame, end to end on the device:

    silk_find_pitch_lags_FIX -> silk_noise_shape_analysis_FIX -> silk_find_pred_coefs_FIX -> silk_process_gains_FIX
        -> silk_prefilter_FIX -> silk_NSQ / silk_NSQ_del_dec -> silk_encode_indices + silk_encode_pulses

i.e. what silk_encode_frame_FIX (opus-fix/silk/fixed/encode_frame_FIX.c:176-317) runs between the VAD and the range coder for
the first pass of a frame. Every stage is one batched kernel of libopusgpu.so driven from a flat record; this module is the
host-side plumbing a batched SILK front end needs between them: after each stage it moves the fields the next records take
from it -- byte ranges of device tensors, no arithmetic -- exactly along the edges that tests/test_silk_chain_cpu.py pins
against the unmodified reference. What the caller provides per frame are the records' remaining fields: the input buffer,
the VAD results, the configuration and the states the previous frame left behind.

Geometry is fixed per chain object (all frames of a batch share fs_kHz / nb_subfr, as one encoder configuration does)."""
import ctypes as C

from . import silk as S


def _off(cls, name):
    d = getattr(cls, name)
    return d.offset, d.size


class SilkAnalysisChain:
    """chain = SilkAnalysisChain(fs_kHz=16, nb_subfr=4); out = chain.run(records...)"""

    def __init__(self, fs_kHz=16, nb_subfr=4):
        if fs_kHz not in (8, 16) or nb_subfr not in (2, 4):
            raise ValueError("fs_kHz 8 or 16, nb_subfr 2 or 4")
        self.fs_kHz, self.nb_subfr = fs_kHz, nb_subfr
        self.frame_length = 5 * fs_kHz * nb_subfr
        self.ltp_mem_length = 20 * fs_kHz

    @staticmethod
    def _move(dst, dcls, dname, src, scls, sname, nbytes=None, src_skip=0):
        do, dn = _off(dcls, dname)
        so, sn = _off(scls, sname)
        n = min(dn, sn - src_skip) if nbytes is None else nbytes
        dst[:, do:do + n] = src[:, so + src_skip:so + src_skip + n]

    @staticmethod
    def _move_i8_to_i32(dst, dcls, dname, src, scls, sname):
        import torch
        do, _ = _off(dcls, dname)
        so, _ = _off(scls, sname)
        v = src[:, so:so + 1].contiguous().view(torch.int8).to(torch.int32)
        dst[:, do:do + 4] = v.view(torch.uint8)

    def run(self, pitch_in, shape_in, fpc_in, gains_in, prefilter_in, prefilter_state, q_in, nsq_state, del_dec, bits_in=None, ec_state=None,
            rate_ctl=None):
        """All arguments are uint8 CUDA tensors [N][record bytes]. shape_in / fpc_in / gains_in / prefilter_in / q_in are
        completed in place from the outputs of the earlier stages; prefilter_state and nsq_state are updated in place.
        q_in is opusgpu_nsq_dd_in when del_dec else opusgpu_nsq_in. With bits_in / ec_state (opusgpu_silk_bits_in with which = 3, the
        frame's range coder) the side information and the excitation are entropy-coded as well (silk_encode_indices +
        silk_encode_pulses): ec_state is updated in place. With rate_ctl (opusgpu_silk_rate_ctl per frame: maxBits, useCBR, condCoding,
        nb_subfr, frame_length set, the rest zero) the bitrate loop of silk_encode_frame_FIX (encode_frame_FIX.c:263-423) runs as well:
        frames over / under their bit budget are quantised and coded again with adjusted gains, up to six more times, each time from
        the coder and quantiser state they entered with; nsq_state / ec_state / "pulses" / "Seed" / bits_in end as the reference
        leaves them, rate_ctl holds LastGainIndex, GainsIndices and the number of passes. Returns a dict of the stage outputs;
        "pulses" is int8 [N][320] (and "Seed" int32 [N] for the delayed-decision quantiser)."""
        import torch
        mv, fl, ltp = self._move, self.frame_length, self.ltp_mem_length
        PO, SI, SO, FI, FO, GI, GO, XI, XO, Q = (S.FindPitchLagsOut, S.NoiseShapeIn, S.NoiseShapeOut, S.FindPredCoefsIn, S.FindPredCoefsOut,
                                                 S.ProcessGainsIn, S.ProcessGainsOut, S.PrefilterIn, S.PrefilterOut, S.NsqIn)
        pitch_out = S.silk_find_pitch_lags(pitch_in)
        # noise_shape_analysis <- find_pitch_lags
        mv(shape_in, SI, "pitch_res", pitch_out, PO, "res", nbytes=2 * fl, src_skip=2 * ltp)
        for name in ("signalType", "LTPCorr_Q15", "predGain_Q16", "pitchL"):
            mv(shape_in, SI, name, pitch_out, PO, name)
        shape_out = S.silk_noise_shape_analysis(shape_in)
        # find_pred_coefs <- find_pitch_lags, noise_shape_analysis
        mv(fpc_in, FI, "res_pitch", pitch_out, PO, "res", nbytes=2 * (ltp + fl))
        mv(fpc_in, FI, "pitchL", pitch_out, PO, "pitchL")
        mv(fpc_in, FI, "signalType", pitch_out, PO, "signalType")
        mv(fpc_in, FI, "Gains_Q16", shape_out, SO, "Gains_Q16")
        mv(fpc_in, FI, "coding_quality_Q14", shape_out, SO, "coding_quality_Q14")
        fpc_out = S.silk_find_pred_coefs(fpc_in)
        # process_gains <- noise_shape_analysis, find_pred_coefs
        mv(gains_in, GI, "Gains_Q16", shape_out, SO, "Gains_Q16")
        for name in ("ResNrg", "ResNrgQ", "LTPredCodGain_Q7"):
            mv(gains_in, GI, name, fpc_out, FO, name)
        for name in ("quantOffsetType", "input_quality_Q14", "coding_quality_Q14"):
            mv(gains_in, GI, name, shape_out, SO, name)
        mv(gains_in, GI, "signalType", pitch_out, PO, "signalType")
        gains_out = S.silk_process_gains(gains_in)
        # prefilter <- noise_shape_analysis, find_pitch_lags
        for name in ("AR1_Q13", "HarmShapeGain_Q14", "HarmBoost_Q14", "Tilt_Q14", "GainsPre_Q14", "LF_shp_Q14", "coding_quality_Q14"):
            mv(prefilter_in, XI, name, shape_out, SO, name)
        mv(prefilter_in, XI, "pitchL", pitch_out, PO, "pitchL")
        mv(prefilter_in, XI, "signalType", pitch_out, PO, "signalType")
        prefilter_out = S.silk_prefilter(prefilter_in, prefilter_state)
        # quantiser <- everything before (opusgpu_nsq_dd_in starts with an opusgpu_nsq_in)
        mv(q_in, Q, "x_Q3", prefilter_out, XO, "xw_Q3")
        for name in ("PredCoef_Q12", "LTPCoef_Q14", "LTP_scale_Q14"):
            mv(q_in, Q, name, fpc_out, FO, name)
        self._move_i8_to_i32(q_in, Q, "NLSFInterpCoef_Q2", fpc_out, FO, "NLSFInterpCoef_Q2")
        for name in ("AR2_Q13", "HarmShapeGain_Q14", "Tilt_Q14", "LF_shp_Q14"):
            mv(q_in, Q, name, shape_out, SO, name)
        for name in ("Gains_Q16", "Lambda_Q10", "quantOffsetType"):
            mv(q_in, Q, name, gains_out, GO, name)
        mv(q_in, Q, "pitchL", pitch_out, PO, "pitchL")
        mv(q_in, Q, "signalType", pitch_out, PO, "signalType")
        out = {"pitch_out": pitch_out, "shape_out": shape_out, "fpc_out": fpc_out, "gains_out": gains_out, "prefilter_out": prefilter_out}
        if rate_ctl is not None:
            if bits_in is None or ec_state is None:
                raise ValueError("rate_ctl needs bits_in and ec_state: the loop measures the coder")
            nsq_entry, ec_entry = nsq_state.clone(), ec_state.clone()       # sNSQ_copy / sRangeEnc_copy (encode_frame_FIX.c:272-273)
        if del_dec:
            dd_out = S.silk_NSQ_del_dec(q_in, nsq_state)
            out["pulses"] = dd_out[:, :320].view(torch.int8)
            out["Seed"] = dd_out[:, 320:324].contiguous().view(torch.int32)[:, 0]
        else:
            out["pulses"] = S.silk_NSQ(q_in, nsq_state)
        if bits_in is not None:
            B, i8to32 = S.SilkBitsIn, self._move_i8_to_i32
            do, _ = _off(B, "pulses")
            bits_in[:, do:do + 320] = out["pulses"].view(torch.uint8)
            mv(bits_in, B, "GainsIndices", gains_out, GO, "GainsIndices")
            mv(bits_in, B, "NLSFIndices", fpc_out, FO, "NLSFIndices")
            mv(bits_in, B, "LTPIndex", fpc_out, FO, "LTPIndex")
            for name in ("NLSFInterpCoef_Q2", "PERIndex", "LTP_scaleIndex"):
                i8to32(bits_in, B, name, fpc_out, FO, name)
            for name in ("lagIndex", "contourIndex", "signalType"):
                mv(bits_in, B, name, pitch_out, PO, name)
            mv(bits_in, B, "quantOffsetType", gains_out, GO, "quantOffsetType")
            if del_dec:                                   # silk_NSQ_del_dec rewrites psIndices->Seed (NSQ_del_dec.c:297)
                so, _ = _off(B, "Seed")
                bits_in[:, so:so + 4] = dd_out[:, 320:324]
            out["bits_out"] = S.silk_encode_bits(bits_in, ec_state)
        if rate_ctl is not None:
            R = S.RateCtl
            for name in ("GainsUnq_Q16", "Gains_Q16", "lastGainIndexPrev", "LastGainIndex", "Lambda_Q10", "GainsIndices"):
                mv(rate_ctl, R, name, gains_out, GO, name)
            self._rate_loop(rate_ctl, q_in, nsq_state, nsq_entry, ec_state, ec_entry, bits_in, del_dec, out)
            out["rate_ctl"] = rate_ctl
        return out

    def _rate_loop(self, rate_ctl, q_in, nsq_state, nsq_entry, ec_state, ec_entry, bits_in, del_dec, out):
        """encode_frame_FIX.c:276-423 over the batch: one opusgpu_silk_rate_control_batch step after every pass decides per frame; the
        frames that need another pass are gathered, quantised + coded from their entry states, and scattered back."""
        import torch
        mv, R, Q, B = self._move, S.RateCtl, S.NsqIn, S.SilkBitsIn
        f0, _ = _off(R, "done")
        po, _ = _off(B, "pulses")
        so, _ = _off(B, "Seed")
        nsq_low = ec_low = None                                             # sNSQ_copy2 / sRangeEnc_copy2 + ec_buf_copy, allocated on first use
        for _ in range(8):                                                  # iter 0 .. maxIter: at most 7 passes, each followed by a step
            S.silk_rate_control(rate_ctl, ec_state)
            flags = rate_ctl[:, f0:f0 + 16].contiguous().view(torch.int32)  # done, recode, save2, restore2
            keep = flags[:, 2].nonzero().squeeze(1)
            if keep.numel():
                if nsq_low is None:
                    nsq_low, ec_low = torch.zeros_like(nsq_state), torch.zeros_like(ec_state)
                nsq_low[keep] = nsq_state[keep]
                ec_low[keep] = ec_state[keep]
            back = flags[:, 3].nonzero().squeeze(1)
            if back.numel():
                nsq_state[back] = nsq_low[back]
                ec_state[back] = ec_low[back]
            rows = flags[:, 1].nonzero().squeeze(1)
            if rows.numel() == 0:
                return
            ctl = rate_ctl[rows]
            q = q_in[rows]
            mv(q, Q, "Gains_Q16", ctl, R, "Gains_Q16")
            mv(q, Q, "Lambda_Q10", ctl, R, "Lambda_Q10")
            nsq, ec, b = nsq_entry[rows], ec_entry[rows], bits_in[rows]
            if del_dec:
                dd_out = S.silk_NSQ_del_dec(q, nsq)
                pulses = dd_out[:, :320]
                b[:, so:so + 4] = dd_out[:, 320:324]
                out["Seed"][rows] = dd_out[:, 320:324].contiguous().view(torch.int32)[:, 0]
            else:
                pulses = S.silk_NSQ(q, nsq).view(torch.uint8)
            b[:, po:po + 320] = pulses
            mv(b, B, "GainsIndices", ctl, R, "GainsIndices")
            bits_out = S.silk_encode_bits(b, ec)
            nsq_state[rows] = nsq
            ec_state[rows] = ec
            bits_in[rows] = b
            q_in[rows] = q
            out["pulses"][rows] = pulses.view(torch.int8)
            out["bits_out"][rows] = bits_out
        raise RuntimeError("silk rate loop: frames still asking for a pass after maxIter")


CHAIN_FED_FIELDS = {       # the record fields run() fills: a caller (and the test) may leave them zero
    "shape_in": (S.NoiseShapeIn, ("pitch_res", "signalType", "LTPCorr_Q15", "predGain_Q16", "pitchL")),
    "fpc_in": (S.FindPredCoefsIn, ("res_pitch", "pitchL", "signalType", "Gains_Q16", "coding_quality_Q14")),
    "gains_in": (S.ProcessGainsIn, ("Gains_Q16", "ResNrg", "ResNrgQ", "LTPredCodGain_Q7", "quantOffsetType", "input_quality_Q14",
                                    "coding_quality_Q14", "signalType")),
    "prefilter_in": (S.PrefilterIn, ("AR1_Q13", "HarmShapeGain_Q14", "HarmBoost_Q14", "Tilt_Q14", "GainsPre_Q14", "LF_shp_Q14",
                                     "coding_quality_Q14", "pitchL", "signalType")),
    "q_in": (S.NsqIn, ("x_Q3", "PredCoef_Q12", "LTPCoef_Q14", "LTP_scale_Q14", "NLSFInterpCoef_Q2", "AR2_Q13", "HarmShapeGain_Q14", "Tilt_Q14",
                       "LF_shp_Q14", "Gains_Q16", "Lambda_Q10", "quantOffsetType", "pitchL", "signalType")),
    "bits_in": (S.SilkBitsIn, ("pulses", "GainsIndices", "NLSFIndices", "LTPIndex", "NLSFInterpCoef_Q2", "PERIndex", "LTP_scaleIndex", "lagIndex",
                               "contourIndex", "signalType", "quantOffsetType")),
}

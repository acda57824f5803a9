"""The SILK analysis chain of one frame, end to end on the device:

    silk_find_pitch_lags_FIX -> silk_noise_shape_analysis_FIX -> silk_find_pred_coefs_FIX -> silk_process_gains_FIX
        -> silk_prefilter_FIX -> silk_NSQ / silk_NSQ_del_dec -> silk_encode_indices + silk_encode_pulses

i.e. what silk_encode_frame_FIX (opus-fix/silk/fixed/encode_frame_FIX.c:176-317) runs between the VAD and the range coder for
the first pass of a frame, and the bitrate loop around quantiser and coder (:263-423) on top of it. The first pass is ONE call of the
C ABI, opusgpu_silk_encode_frames_batch (concentus_amd/csrc/silk_chain.hip): the eight batched kernels back to back and, between
them, small kernels that complete the next records from the outputs of the earlier stages -- field to field, no arithmetic, exactly
along the edges that tests/test_silk_chain_cpu.py pins against the unmodified reference. What the caller provides per frame are the
records' remaining fields: the input buffer, the VAD results, the configuration and the states the previous frame left behind. With
run(..., rate_ctl=...) the loop runs as well, also inside ONE call of the C ABI (opusgpu_silk_encode_frames_cbr_batch): the frames that
ask for another pass are listed on the device and quantiser + coder run over that list in place; nothing here touches a record.

Geometry is fixed per chain object (all frames of a batch share fs_kHz / nb_subfr, as one encoder configuration does)."""
import ctypes as C

from . import silk as S


class ChainBufs(C.Structure):
    """opusgpu_silk_chain_bufs (include/opusgpu_silk.h): device pointers to the n records of every stage."""
    _fields_ = [(k, C.c_void_p) for k in ("pitch_in", "pitch_out", "shape_in", "shape_out", "fpc_in", "fpc_out", "gains_in", "gains_out",
                                          "prefilter_in", "prefilter_state", "prefilter_out", "q_in", "nsq_state", "q_out", "bits_in", "ec_state",
                                          "bits_out", "workspace")] + [("workspace_bytes", C.c_size_t)]


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
            rate_ctl=None, streams=None, frame_input=None):
        """All arguments are uint8 CUDA tensors [N][record bytes]. shape_in / fpc_in / gains_in / prefilter_in / q_in are
        completed in place from the outputs of the earlier stages; prefilter_state and nsq_state are updated in place.
        q_in is opusgpu_nsq_dd_in when del_dec else opusgpu_nsq_in. With bits_in / ec_state (opusgpu_silk_bits_in with which = 3, the
        frame's range coder) the side information and the excitation are entropy-coded as well (silk_encode_indices +
        silk_encode_pulses): ec_state is updated in place. With rate_ctl (opusgpu_silk_rate_ctl per frame: maxBits, useCBR, condCoding,
        nb_subfr, frame_length set, the rest zero) the bitrate loop of silk_encode_frame_FIX (encode_frame_FIX.c:263-423) runs as well:
        frames over / under their bit budget are quantised and coded again with adjusted gains, up to six more times, each time from
        the coder and quantiser state they entered with; nsq_state / ec_state / "pulses" / "Seed" / bits_in end as the reference
        leaves them, rate_ctl holds LastGainIndex, GainsIndices and the number of passes. STREAMS MODE: with streams (uint8 [N][sizeof
        opusgpu_silk_stream], one record per stream, updated in place) and frame_input (int16 [N][320]: the frame's samples) the fields a
        frame inherits from the previous one (CARRIED_FIELDS) are filled on the device before the frame and the stream records updated
        after it (opusgpu_silk_stream_carry_in / _out); prefilter_state / nsq_state are then the streams' own. Returns a dict of the stage outputs;
        "pulses" is int8 [N][320] (and "Seed" int32 [N] for the delayed-decision quantiser)."""
        import torch
        from . import lib as _lib
        mv = self._move
        GO = S.ProcessGainsOut
        n, dev = pitch_in.shape[0], pitch_in.device
        recs = (("pitch_in", pitch_in, "find_pitch_lags_in"), ("shape_in", shape_in, "noise_shape_in"), ("fpc_in", fpc_in, "find_pred_coefs_in"),
                ("gains_in", gains_in, "process_gains_in"), ("prefilter_in", prefilter_in, "prefilter_in"),
                ("prefilter_state", prefilter_state, "prefilter_state"), ("q_in", q_in, "nsq_dd_in" if del_dec else "nsq_in"),
                ("nsq_state", nsq_state, "nsq_state")) + ((("bits_in", bits_in, "silk_bits_in"), ("ec_state", ec_state, "ec_state")) if bits_in is not None else ())
        for name, t, size in recs:
            S._check(t, S.SIZES[size], name)
            if t.shape[0] != n:
                raise ValueError("%s: %d records for %d frames" % (name, t.shape[0], n))
        if rate_ctl is not None and (bits_in is None or ec_state is None):
            raise ValueError("rate_ctl needs bits_in and ec_state: the loop measures the coder")

        def new(size, zero=False):
            return (torch.zeros if zero else torch.empty)((n, S.SIZES[size]), dtype=torch.uint8, device=dev)
        pitch_out, shape_out, fpc_out, gains_out, prefilter_out = (new(k) for k in ("find_pitch_lags_out", "noise_shape_out", "find_pred_coefs_out",
                                                                                   "process_gains_out", "prefilter_out"))
        q_out = new("nsq_dd_out" if del_dec else "nsq_out", zero=True)
        bits_out = new("silk_bits_out") if bits_in is not None else None
        L = _lib.load()
        need = (L.opusgpu_silk_nsq_del_dec_workspace_bytes if del_dec else L.opusgpu_silk_nsq_workspace_bytes)(n)
        ws = S._scratch(dev, need, "nsq_del_dec" if del_dec else "nsq")
        # one C call: the eight kernels and the field moves between their records (csrc/silk_chain.hip)
        bufs = ChainBufs(*[C.c_void_p(t.data_ptr() if t is not None else None) for t in (
            pitch_in, pitch_out, shape_in, shape_out, fpc_in, fpc_out, gains_in, gains_out, prefilter_in, prefilter_state, prefilter_out,
            q_in, nsq_state, q_out, bits_in, ec_state if bits_in is not None else None, bits_out, ws)], C.c_size_t(ws.numel()))
        if streams is not None:
            S._check(streams, S.SIZES["silk_stream"], "streams")
            if not (frame_input is not None and frame_input.is_cuda and frame_input.dtype == torch.int16 and frame_input.is_contiguous()
                    and tuple(frame_input.shape) == (n, 320) and streams.shape[0] == n):
                raise ValueError("frame_input must be a contiguous int16 CUDA tensor [streams][320], one row per stream record")
            _lib.check(L.opusgpu_silk_stream_carry_in(C.c_void_p(streams.data_ptr()), C.c_void_p(frame_input.data_ptr()), C.byref(bufs),
                                                      self.fs_kHz, self.nb_subfr, 1 if del_dec else 0, n, _lib.current_stream_handle()),
                       "opusgpu_silk_stream_carry_in")
        if rate_ctl is None:
            rc = L.opusgpu_silk_encode_frames_batch(C.byref(bufs), self.fs_kHz, self.nb_subfr, 1 if del_dec else 0, n, _lib.current_stream_handle())
            _lib.check(rc, "opusgpu_silk_encode_frames_batch")
        else:
            # the whole of silk_encode_frame_FIX, loop included, in one call of the C ABI (csrc/silk_chain.hip)
            S._check(rate_ctl, S.SIZES["silk_rate_ctl"], "rate_ctl")
            lws = S._scratch(dev, L.opusgpu_silk_encode_frames_cbr_workspace_bytes(n), "silk_cbr_loop")
            passes = C.c_int(0)
            rc = L.opusgpu_silk_encode_frames_cbr_batch(C.byref(bufs), C.c_void_p(rate_ctl.data_ptr()), self.fs_kHz, self.nb_subfr,
                                                        1 if del_dec else 0, n, C.c_void_p(lws.data_ptr()), C.c_size_t(lws.numel()),
                                                        C.byref(passes), _lib.current_stream_handle())
            _lib.check(rc, "opusgpu_silk_encode_frames_cbr_batch")
            self.last_loop_iterations = passes.value
        if streams is not None:
            _lib.check(L.opusgpu_silk_stream_carry_out(C.c_void_p(streams.data_ptr()), C.byref(bufs),
                                                       C.c_void_p(rate_ctl.data_ptr()) if rate_ctl is not None else None,
                                                       self.fs_kHz, self.nb_subfr, n, _lib.current_stream_handle()),
                       "opusgpu_silk_stream_carry_out")
        out = {"pitch_out": pitch_out, "shape_out": shape_out, "fpc_out": fpc_out, "gains_out": gains_out, "prefilter_out": prefilter_out,
               "pulses": q_out[:, :320].view(torch.int8)}
        if del_dec:
            out["Seed"] = q_out[:, 320:324].contiguous().view(torch.int32)[:, 0]
        if bits_in is not None:
            out["bits_out"] = bits_out
        if rate_ctl is not None:
            out["rate_ctl"] = rate_ctl
        # a record that fails the device-side checks yields zeroed outputs and is counted -- say so instead of
        # returning it like a good frame
        bad = S.bad_records()
        if bad:
            raise _lib.OpusGpuError(-1, "SilkAnalysisChain.run: %d record(s) failed the device-side checks (status fields of the *_out records)" % bad)
        return out


CARRIED_FIELDS = {         # streams mode: the record fields opusgpu_silk_stream_carry_in fills from the stream record + the frame's samples
    "pitch_in": (S.FindPitchLagsIn, ("x_buf", "prevLag", "prevSignalType", "first_frame_after_reset", "LTPCorr_Q15")),
    "shape_in": (S.NoiseShapeIn, ("x", "HarmBoost_smth_Q16", "HarmShapeGain_smth_Q16", "Tilt_smth_Q16")),
    "fpc_in": (S.FindPredCoefsIn, ("x", "prev_NLSFq_Q15", "first_frame_after_reset", "sum_log_gain_Q7")),
    "gains_in": (S.ProcessGainsIn, ("LastGainIndex",)),
    "prefilter_in": (S.PrefilterIn, ("x",)),
    "q_in": (S.NsqIn, ("Seed",)),
    "bits_in": (S.SilkBitsIn, ("Seed", "ec_prevSignalType", "ec_prevLagIndex")),
}

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

"""Host-side mirror of the two SILK operators on the hot path (BASELINE config #4):
silk_burg_modified() (opus-fix/silk/SigProc_FIX.h:601, fixed/burg_modified_FIX.c:45) and silk_NSQ()
(opus-fix/silk/main.h:245-268, NSQ.c:74), over batches of function-boundary records held in device
tensors (byte tensors with the layouts of include/opusgpu_silk.h)."""
import ctypes as C

from . import lib as _lib


class BurgIn(C.Structure):
    _fields_ = [("x", C.c_int16 * 384), ("minInvGain_Q30", C.c_int32), ("subfr_length", C.c_int32),
                ("nb_subfr", C.c_int32), ("D", C.c_int32)]


class BurgOut(C.Structure):
    _fields_ = [("res_nrg", C.c_int32), ("res_nrg_Q", C.c_int32), ("A_Q16", C.c_int32 * 16)]


class NsqState(C.Structure):
    _fields_ = [("xq", C.c_int16 * 640), ("sLTP_shp_Q14", C.c_int32 * 640), ("sLPC_Q14", C.c_int32 * 112),
                ("sAR2_Q14", C.c_int32 * 16), ("sLF_AR_shp_Q14", C.c_int32), ("lagPrev", C.c_int32),
                ("sLTP_buf_idx", C.c_int32), ("sLTP_shp_buf_idx", C.c_int32), ("rand_seed", C.c_int32),
                ("prev_gain_Q16", C.c_int32), ("rewhite_flag", C.c_int32)]


class NsqIn(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("nb_subfr", "subfr_length", "frame_length", "ltp_mem_length", "predictLPCOrder",
                                         "shapingLPCOrder", "signalType", "quantOffsetType", "NLSFInterpCoef_Q2", "Seed",
                                         "Lambda_Q10", "LTP_scale_Q14")] + \
               [("HarmShapeGain_Q14", C.c_int32 * 4), ("Tilt_Q14", C.c_int32 * 4), ("LF_shp_Q14", C.c_int32 * 4),
                ("Gains_Q16", C.c_int32 * 4), ("pitchL", C.c_int32 * 4), ("x_Q3", C.c_int32 * 320),
                ("PredCoef_Q12", C.c_int16 * 32), ("LTPCoef_Q14", C.c_int16 * 20), ("AR2_Q13", C.c_int16 * 64)]


SIZES = {"burg_in": C.sizeof(BurgIn), "burg_out": C.sizeof(BurgOut), "nsq_in": C.sizeof(NsqIn),
         "nsq_state": C.sizeof(NsqState), "nsq_out": 320}


def _check(t, rec_bytes, name):
    import torch
    if not (t.is_cuda and t.dtype == torch.uint8 and t.is_contiguous() and t.dim() == 2 and t.shape[1] == rec_bytes):
        raise ValueError("%s must be a contiguous uint8 CUDA tensor [records][%d]" % (name, rec_bytes))


def silk_burg_modified(burg_in, burg_out=None):
    """burg_in uint8 [N][784] (opusgpu_burg_in records) -> uint8 [N][72] (opusgpu_burg_out)."""
    import torch
    _check(burg_in, SIZES["burg_in"], "burg_in")
    n = burg_in.shape[0]
    if burg_out is None:
        burg_out = torch.empty((n, SIZES["burg_out"]), dtype=torch.uint8, device=burg_in.device)
    _check(burg_out, SIZES["burg_out"], "burg_out")
    rc = _lib.load().opusgpu_silk_burg_modified_batch(burg_in.data_ptr(), burg_out.data_ptr(), n, _lib.current_stream_handle())
    _lib.check(rc, "opusgpu_silk_burg_modified_batch")
    return burg_out


_WS = {}


def _scratch(device, need, op):
    """Re-whitening scratch of the NSQ kernels, per (device, stream, operator): calls issued on different HIP streams
    (bench.py's mixed workload runs SILK on a side stream) must not share it, and neither may the two quantisers."""
    import torch
    key = (device, _lib.current_stream_handle().value, op)
    ws = _WS.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty((max(need, 1),), dtype=torch.uint8, device=device)
        _WS[key] = ws
    return ws


def bad_records():
    """Records the SILK batch kernels skipped on the current device since the last call because their header failed
    the device-side bounds checks (include/opusgpu_silk.h); waits for the current stream."""
    return _lib.check(_lib.load().opusgpu_silk_bad_records(_lib.current_stream_handle()), "opusgpu_silk_bad_records")


def silk_NSQ(nsq_in, nsq_state, pulses=None):
    """nsq_in uint8 [N][1640], nsq_state uint8 [N][4380] (updated in place) -> pulses int8 [N][320]."""
    import torch
    _check(nsq_in, SIZES["nsq_in"], "nsq_in")
    _check(nsq_state, SIZES["nsq_state"], "nsq_state")
    n = nsq_in.shape[0]
    if pulses is None:
        pulses = torch.zeros((n, 320), dtype=torch.int8, device=nsq_in.device)
    if not (pulses.is_cuda and pulses.dtype in (torch.int8, torch.uint8) and pulses.is_contiguous() and tuple(pulses.shape) == (n, 320)):
        raise ValueError("pulses must be a contiguous int8 CUDA tensor [records][320]")
    L = _lib.load()
    need = L.opusgpu_silk_nsq_workspace_bytes(n)
    ws = _scratch(nsq_in.device, need, "nsq")
    rc = L.opusgpu_silk_nsq_batch(nsq_in.data_ptr(), nsq_state.data_ptr(), pulses.data_ptr(), n, ws.data_ptr(), ws.numel(),
                                  _lib.current_stream_handle())
    _lib.check(rc, "opusgpu_silk_nsq_batch")
    return pulses


class NsqDdIn(C.Structure):
    """opusgpu_nsq_dd_in: the silk_NSQ() arguments + nStatesDelayedDecision, warping_Q16."""
    _fields_ = [("base", NsqIn), ("nStatesDelayedDecision", C.c_int32), ("warping_Q16", C.c_int32)]


SIZES["nsq_dd_in"] = C.sizeof(NsqDdIn)
SIZES["nsq_dd_out"] = 324


def silk_NSQ_del_dec(dd_in, nsq_state, dd_out=None):
    """silk_NSQ_del_dec() (opus-fix/silk/main.h:271-296, NSQ_del_dec.c:112) over a batch of records:
    dd_in uint8 [N][1648], nsq_state uint8 [N][4380] (updated in place) -> dd_out uint8 [N][324] (pulses int8[320], Seed)."""
    import torch
    _check(dd_in, SIZES["nsq_dd_in"], "dd_in")
    _check(nsq_state, SIZES["nsq_state"], "nsq_state")
    n = dd_in.shape[0]
    if dd_out is None:
        dd_out = torch.zeros((n, SIZES["nsq_dd_out"]), dtype=torch.uint8, device=dd_in.device)
    _check(dd_out, SIZES["nsq_dd_out"], "dd_out")
    L = _lib.load()
    need = L.opusgpu_silk_nsq_del_dec_workspace_bytes(n)
    ws = _scratch(dd_in.device, need, "nsq_del_dec")
    rc = L.opusgpu_silk_nsq_del_dec_batch(dd_in.data_ptr(), nsq_state.data_ptr(), dd_out.data_ptr(), n, ws.data_ptr(), ws.numel(),
                                          _lib.current_stream_handle())
    _lib.check(rc, "opusgpu_silk_nsq_del_dec_batch")
    return dd_out


class FindLpcIn(C.Structure):
    """opusgpu_find_lpc_in: one silk_find_LPC_FIX() call (opus-fix/silk/fixed/find_LPC_FIX.c:37)."""
    _fields_ = [("x", C.c_int16 * 384), ("minInvGain_Q30", C.c_int32), ("subfr_length", C.c_int32), ("nb_subfr", C.c_int32),
                ("predictLPCOrder", C.c_int32), ("useInterpolatedNLSFs", C.c_int32), ("first_frame_after_reset", C.c_int32),
                ("prev_NLSFq_Q15", C.c_int16 * 16), ("reserved", C.c_int32 * 2)]


class FindLpcOut(C.Structure):
    _fields_ = [("NLSF_Q15", C.c_int16 * 16), ("NLSFInterpCoef_Q2", C.c_int32), ("status", C.c_int32)]


SIZES["find_lpc_in"] = C.sizeof(FindLpcIn)
SIZES["find_lpc_out"] = C.sizeof(FindLpcOut)


def silk_find_LPC(lpc_in, lpc_out=None):
    """silk_find_LPC_FIX() over a batch of records: lpc_in uint8 [N][832] (opusgpu_find_lpc_in) -> uint8 [N][40]
    (NLSF_Q15 int16[16], NLSFInterpCoef_Q2, status)."""
    import torch
    _check(lpc_in, SIZES["find_lpc_in"], "lpc_in")
    n = lpc_in.shape[0]
    if lpc_out is None:
        lpc_out = torch.empty((n, SIZES["find_lpc_out"]), dtype=torch.uint8, device=lpc_in.device)
    _check(lpc_out, SIZES["find_lpc_out"], "lpc_out")
    rc = _lib.load().opusgpu_silk_find_lpc_batch(lpc_in.data_ptr(), lpc_out.data_ptr(), n, _lib.current_stream_handle())
    _lib.check(rc, "opusgpu_silk_find_lpc_batch")
    return lpc_out


class ProcessNlsfIn(C.Structure):
    """opusgpu_process_nlsf_in: one silk_process_NLSFs() call (opus-fix/silk/process_NLSFs.c:35)."""
    _fields_ = [("NLSF_Q15", C.c_int16 * 16), ("prev_NLSFq_Q15", C.c_int16 * 16), ("speech_activity_Q8", C.c_int32),
                ("nb_subfr", C.c_int32), ("predictLPCOrder", C.c_int32), ("useInterpolatedNLSFs", C.c_int32),
                ("NLSFInterpCoef_Q2", C.c_int32), ("NLSF_MSVQ_Survivors", C.c_int32), ("signalType", C.c_int32), ("reserved", C.c_int32)]


class ProcessNlsfOut(C.Structure):
    _fields_ = [("PredCoef_Q12", C.c_int16 * 32), ("NLSF_Q15", C.c_int16 * 16), ("NLSFIndices", C.c_int8 * 17), ("pad", C.c_int8 * 3),
                ("status", C.c_int32)]


class ResNrgIn(C.Structure):
    """opusgpu_res_nrg_in: one silk_residual_energy_FIX() call (opus-fix/silk/fixed/residual_energy_FIX.c:37)."""
    _fields_ = [("x", C.c_int16 * 384), ("a_Q12", C.c_int16 * 32), ("gains", C.c_int32 * 4), ("subfr_length", C.c_int32),
                ("nb_subfr", C.c_int32), ("LPC_order", C.c_int32), ("reserved", C.c_int32)]


class ResNrgOut(C.Structure):
    _fields_ = [("nrgs", C.c_int32 * 4), ("nrgsQ", C.c_int32 * 4), ("status", C.c_int32), ("reserved", C.c_int32)]


SIZES["process_nlsf_in"] = C.sizeof(ProcessNlsfIn)
SIZES["process_nlsf_out"] = C.sizeof(ProcessNlsfOut)
SIZES["res_nrg_in"] = C.sizeof(ResNrgIn)
SIZES["res_nrg_out"] = C.sizeof(ResNrgOut)


def _record_op(symbol, rec_in, rec_out, size_in, size_out, name):
    import torch
    _check(rec_in, size_in, name + "_in")
    n = rec_in.shape[0]
    if rec_out is None:
        rec_out = torch.empty((n, size_out), dtype=torch.uint8, device=rec_in.device)
    _check(rec_out, size_out, name + "_out")
    rc = getattr(_lib.load(), symbol)(rec_in.data_ptr(), rec_out.data_ptr(), n, _lib.current_stream_handle())
    _lib.check(rc, symbol)
    return rec_out


def silk_process_NLSFs(nlsf_in, nlsf_out=None):
    """silk_process_NLSFs() over a batch of records: nlsf_in uint8 [N][96] (opusgpu_process_nlsf_in) -> uint8 [N][120]
    (PredCoef_Q12 int16[2][16], quantised NLSF_Q15 int16[16], NLSFIndices int8[17], status)."""
    return _record_op("opusgpu_silk_process_nlsfs_batch", nlsf_in, nlsf_out, SIZES["process_nlsf_in"], SIZES["process_nlsf_out"],
                      "nlsf")


def silk_residual_energy(nrg_in, nrg_out=None):
    """silk_residual_energy_FIX() over a batch of records: nrg_in uint8 [N][864] (opusgpu_res_nrg_in) -> uint8 [N][40]
    (nrgs int32[4], nrgsQ int32[4], status)."""
    return _record_op("opusgpu_silk_residual_energy_batch", nrg_in, nrg_out, SIZES["res_nrg_in"], SIZES["res_nrg_out"], "res_nrg")


class FindPredCoefsIn(C.Structure):
    """opusgpu_find_pred_coefs_in: one silk_find_pred_coefs_FIX() call (opus-fix/silk/fixed/find_pred_coefs_FIX.c:35)."""
    _fields_ = [("res_pitch", C.c_int16 * 640), ("x", C.c_int16 * 640), ("Gains_Q16", C.c_int32 * 4), ("pitchL", C.c_int32 * 4),
                ("prev_NLSFq_Q15", C.c_int16 * 16)] + [(k, C.c_int32) for k in (
                    "nb_subfr", "subfr_length", "predictLPCOrder", "ltp_mem_length", "signalType", "condCoding",
                    "first_frame_after_reset", "useInterpolatedNLSFs", "speech_activity_Q8", "NLSF_MSVQ_Survivors", "mu_LTP_Q9",
                    "LTPQuantLowComplexity", "sum_log_gain_Q7", "coding_quality_Q14", "PacketLoss_perc", "nFramesPerPacket")]


class FindPredCoefsOut(C.Structure):
    _fields_ = [("PredCoef_Q12", C.c_int16 * 32), ("LTPCoef_Q14", C.c_int16 * 20), ("NLSF_Q15", C.c_int16 * 16), ("ResNrg", C.c_int32 * 4),
                ("ResNrgQ", C.c_int32 * 4), ("LTPredCodGain_Q7", C.c_int32), ("LTP_scale_Q14", C.c_int32), ("sum_log_gain_Q7", C.c_int32),
                ("NLSFIndices", C.c_int8 * 17), ("NLSFInterpCoef_Q2", C.c_int8), ("LTPIndex", C.c_int8 * 4), ("PERIndex", C.c_int8),
                ("LTP_scaleIndex", C.c_int8), ("status", C.c_int32)]


SIZES["find_pred_coefs_in"] = C.sizeof(FindPredCoefsIn)
SIZES["find_pred_coefs_out"] = C.sizeof(FindPredCoefsOut)


def silk_find_pred_coefs(fpc_in, fpc_out=None):
    """silk_find_pred_coefs_FIX() over a batch of records: fpc_in uint8 [N][2688] (opusgpu_find_pred_coefs_in) -> uint8 [N][208]
    (opusgpu_find_pred_coefs_out: every field the call writes in psEnc / psEncCtrl, then status)."""
    return _record_op("opusgpu_silk_find_pred_coefs_batch", fpc_in, fpc_out, SIZES["find_pred_coefs_in"], SIZES["find_pred_coefs_out"],
                      "fpc")


class ProcessGainsIn(C.Structure):
    """opusgpu_process_gains_in: one silk_process_gains_FIX() call (opus-fix/silk/fixed/process_gains_FIX.c:37)."""
    _fields_ = [("Gains_Q16", C.c_int32 * 4), ("ResNrg", C.c_int32 * 4), ("ResNrgQ", C.c_int32 * 4)] + [(k, C.c_int32) for k in (
        "LTPredCodGain_Q7", "signalType", "nb_subfr", "subfr_length", "SNR_dB_Q7", "LastGainIndex", "condCoding", "input_tilt_Q15",
        "quantOffsetType", "nStatesDelayedDecision", "speech_activity_Q8", "input_quality_Q14", "coding_quality_Q14")] + [
        ("reserved", C.c_int32 * 3)]


class ProcessGainsOut(C.Structure):
    _fields_ = [("Gains_Q16", C.c_int32 * 4), ("GainsUnq_Q16", C.c_int32 * 4), ("Lambda_Q10", C.c_int32), ("LastGainIndex", C.c_int32),
                ("lastGainIndexPrev", C.c_int32), ("quantOffsetType", C.c_int32), ("GainsIndices", C.c_int8 * 4), ("status", C.c_int32)]


SIZES["process_gains_in"] = C.sizeof(ProcessGainsIn)
SIZES["process_gains_out"] = C.sizeof(ProcessGainsOut)


def silk_process_gains(gains_in, gains_out=None):
    """silk_process_gains_FIX() over a batch of records: gains_in uint8 [N][112] (opusgpu_process_gains_in) -> uint8 [N][56]
    (opusgpu_process_gains_out)."""
    return _record_op("opusgpu_silk_process_gains_batch", gains_in, gains_out, SIZES["process_gains_in"], SIZES["process_gains_out"],
                      "gains")


class NoiseShapeIn(C.Structure):
    """opusgpu_noise_shape_in: one silk_noise_shape_analysis_FIX() call (opus-fix/silk/fixed/noise_shape_analysis_FIX.c:146)."""
    _fields_ = [("x", C.c_int16 * 480), ("pitch_res", C.c_int16 * 320)] + [(k, C.c_int32) for k in (
        "fs_kHz", "nb_subfr", "subfr_length", "la_shape", "shapeWinLength", "shapingLPCOrder", "warping_Q16", "SNR_dB_Q7", "useCBR",
        "speech_activity_Q8", "signalType", "LTPCorr_Q15")] + [("input_quality_bands_Q15", C.c_int32 * 2), ("predGain_Q16", C.c_int32),
        ("reserved", C.c_int32), ("pitchL", C.c_int32 * 4), ("HarmBoost_smth_Q16", C.c_int32), ("HarmShapeGain_smth_Q16", C.c_int32),
        ("Tilt_smth_Q16", C.c_int32), ("reserved2", C.c_int32)]


class NoiseShapeOut(C.Structure):
    _fields_ = [("Gains_Q16", C.c_int32 * 4), ("GainsPre_Q14", C.c_int32 * 4), ("AR1_Q13", C.c_int16 * 64), ("AR2_Q13", C.c_int16 * 64),
                ("LF_shp_Q14", C.c_int32 * 4), ("HarmBoost_Q14", C.c_int32 * 4), ("HarmShapeGain_Q14", C.c_int32 * 4),
                ("Tilt_Q14", C.c_int32 * 4), ("HarmBoost_smth_Q16", C.c_int32), ("HarmShapeGain_smth_Q16", C.c_int32),
                ("Tilt_smth_Q16", C.c_int32), ("input_quality_Q14", C.c_int32), ("coding_quality_Q14", C.c_int32),
                ("sparseness_Q8", C.c_int32), ("quantOffsetType", C.c_int32), ("status", C.c_int32)]


SIZES["noise_shape_in"] = C.sizeof(NoiseShapeIn)
SIZES["noise_shape_out"] = C.sizeof(NoiseShapeOut)


def silk_noise_shape_analysis(shape_in, shape_out=None):
    """silk_noise_shape_analysis_FIX() over a batch of records: shape_in uint8 [N][1696] (opusgpu_noise_shape_in) -> uint8 [N][384]
    (opusgpu_noise_shape_out)."""
    return _record_op("opusgpu_silk_noise_shape_analysis_batch", shape_in, shape_out, SIZES["noise_shape_in"], SIZES["noise_shape_out"],
                      "shape")


class PrefilterState(C.Structure):
    """opusgpu_prefilter_state == silk_prefilter_state_FIX (opus-fix/silk/fixed/structs_FIX.h:53-62)."""
    _fields_ = [("sLTP_shp", C.c_int16 * 512), ("sAR_shp", C.c_int32 * 17), ("sLTP_shp_buf_idx", C.c_int32), ("sLF_AR_shp_Q12", C.c_int32),
                ("sLF_MA_shp_Q12", C.c_int32), ("sHarmHP_Q2", C.c_int32), ("rand_seed", C.c_int32), ("lagPrev", C.c_int32)]


class PrefilterIn(C.Structure):
    """opusgpu_prefilter_in: one silk_prefilter_FIX() call (opus-fix/silk/fixed/prefilter_FIX.c:102)."""
    _fields_ = [("x", C.c_int16 * 320), ("AR1_Q13", C.c_int16 * 64), ("pitchL", C.c_int32 * 4), ("HarmShapeGain_Q14", C.c_int32 * 4),
                ("HarmBoost_Q14", C.c_int32 * 4), ("Tilt_Q14", C.c_int32 * 4), ("GainsPre_Q14", C.c_int32 * 4), ("LF_shp_Q14", C.c_int32 * 4),
                ("coding_quality_Q14", C.c_int32), ("nb_subfr", C.c_int32), ("subfr_length", C.c_int32), ("signalType", C.c_int32),
                ("warping_Q16", C.c_int32), ("shapingLPCOrder", C.c_int32), ("reserved", C.c_int32 * 2)]


class PrefilterOut(C.Structure):
    _fields_ = [("xw_Q3", C.c_int32 * 320), ("status", C.c_int32), ("reserved", C.c_int32 * 3)]


SIZES["prefilter_in"] = C.sizeof(PrefilterIn)
SIZES["prefilter_state"] = C.sizeof(PrefilterState)
SIZES["prefilter_out"] = C.sizeof(PrefilterOut)


def silk_prefilter(pf_in, pf_state, pf_out=None):
    """silk_prefilter_FIX() over a batch of records: pf_in uint8 [N][896], pf_state uint8 [N][1116] (silk_prefilter_state_FIX, updated
    in place) -> uint8 [N][1296] (xw_Q3 int32[320], status)."""
    import torch
    _check(pf_in, SIZES["prefilter_in"], "prefilter_in")
    _check(pf_state, SIZES["prefilter_state"], "prefilter_state")
    n = pf_in.shape[0]
    if pf_state.shape[0] != n:
        raise ValueError("prefilter_state: %d records for %d inputs" % (pf_state.shape[0], n))
    if pf_out is None:
        pf_out = torch.empty((n, SIZES["prefilter_out"]), dtype=torch.uint8, device=pf_in.device)
    _check(pf_out, SIZES["prefilter_out"], "prefilter_out")
    rc = _lib.load().opusgpu_silk_prefilter_batch(pf_in.data_ptr(), pf_state.data_ptr(), pf_out.data_ptr(), n, _lib.current_stream_handle())
    _lib.check(rc, "opusgpu_silk_prefilter_batch")
    return pf_out


class FindPitchLagsIn(C.Structure):
    """opusgpu_find_pitch_lags_in: one silk_find_pitch_lags_FIX() call (opus-fix/silk/fixed/find_pitch_lags_FIX.c:37)."""
    _fields_ = [("x_buf", C.c_int16 * 672)] + [(k, C.c_int32) for k in (
        "fs_kHz", "nb_subfr", "frame_length", "ltp_mem_length", "la_pitch", "pitch_LPC_win_length", "pitchEstimationLPCOrder",
        "pitchEstimationComplexity", "pitchEstimationThreshold_Q16", "signalType", "first_frame_after_reset", "speech_activity_Q8",
        "prevSignalType", "input_tilt_Q15", "prevLag", "LTPCorr_Q15")]


class FindPitchLagsOut(C.Structure):
    _fields_ = [("res", C.c_int16 * 672), ("pitchL", C.c_int32 * 4), ("lagIndex", C.c_int32), ("contourIndex", C.c_int32),
                ("LTPCorr_Q15", C.c_int32), ("signalType", C.c_int32), ("predGain_Q16", C.c_int32), ("status", C.c_int32),
                ("reserved", C.c_int32 * 2)]


SIZES["find_pitch_lags_in"] = C.sizeof(FindPitchLagsIn)
SIZES["find_pitch_lags_out"] = C.sizeof(FindPitchLagsOut)


def silk_find_pitch_lags(pitch_in, pitch_out=None):
    """silk_find_pitch_lags_FIX() over a batch of records: pitch_in uint8 [N][1408] (opusgpu_find_pitch_lags_in) -> uint8 [N][1392]
    (res int16[672], pitchL, lagIndex, contourIndex, LTPCorr_Q15, signalType, predGain_Q16, status)."""
    return _record_op("opusgpu_silk_find_pitch_lags_batch", pitch_in, pitch_out, SIZES["find_pitch_lags_in"], SIZES["find_pitch_lags_out"],
                      "pitch")


class EcState(C.Structure):
    """opusgpu_ec_state: the reference's ec_ctx (opus-fix/celt/entcode.h:63-94) as a record, buffer included."""
    _fields_ = [("storage", C.c_uint32), ("end_offs", C.c_uint32), ("end_window", C.c_uint32), ("nend_bits", C.c_int32), ("nbits_total", C.c_int32),
                ("offs", C.c_uint32), ("rng", C.c_uint32), ("val", C.c_uint32), ("ext", C.c_uint32), ("rem", C.c_int32), ("error", C.c_int32),
                ("reserved", C.c_uint32), ("buf", C.c_uint8 * 1280)]


class SilkBitsIn(C.Structure):
    """opusgpu_silk_bits_in: one silk_encode_indices() and / or silk_encode_pulses() call (opus-fix/silk/encode_indices.c:36,
    silk/encode_pulses.c:64)."""
    _fields_ = [("pulses", C.c_int8 * 320), ("GainsIndices", C.c_int8 * 4), ("LTPIndex", C.c_int8 * 4), ("NLSFIndices", C.c_int8 * 17),
                ("pad", C.c_int8 * 3)] + [(k, C.c_int32) for k in (
                    "lagIndex", "contourIndex", "signalType", "quantOffsetType", "NLSFInterpCoef_Q2", "PERIndex", "LTP_scaleIndex", "Seed",
                    "nb_subfr", "fs_kHz", "predictLPCOrder", "frame_length", "condCoding", "ec_prevSignalType", "ec_prevLagIndex", "which",
                    "reserved")]


class SilkBitsOut(C.Structure):
    _fields_ = [("ec_prevSignalType", C.c_int32), ("ec_prevLagIndex", C.c_int32), ("status", C.c_int32), ("reserved", C.c_int32)]


SIZES["silk_bits_in"] = C.sizeof(SilkBitsIn)
SIZES["ec_state"] = C.sizeof(EcState)
SIZES["silk_bits_out"] = C.sizeof(SilkBitsOut)


def silk_encode_bits(bits_in, ec_state, bits_out=None):
    """silk_encode_indices() / silk_encode_pulses() over a batch of records: bits_in uint8 [N][416], ec_state uint8 [N][1328] (the range
    coder of each frame, updated in place: fields and written bytes) -> uint8 [N][16] (ec_prevSignalType, ec_prevLagIndex, status)."""
    import torch
    _check(bits_in, SIZES["silk_bits_in"], "bits_in")
    _check(ec_state, SIZES["ec_state"], "ec_state")
    n = bits_in.shape[0]
    if ec_state.shape[0] != n:
        raise ValueError("ec_state: %d records for %d inputs" % (ec_state.shape[0], n))
    if bits_out is None:
        bits_out = torch.empty((n, SIZES["silk_bits_out"]), dtype=torch.uint8, device=bits_in.device)
    _check(bits_out, SIZES["silk_bits_out"], "bits_out")
    rc = _lib.load().opusgpu_silk_encode_bits_batch(bits_in.data_ptr(), ec_state.data_ptr(), bits_out.data_ptr(), n, _lib.current_stream_handle())
    _lib.check(rc, "opusgpu_silk_encode_bits_batch")
    return bits_out


class RateCtl(C.Structure):
    """opusgpu_silk_rate_ctl: the locals of silk_encode_frame_FIX's bitrate loop for one frame (opus-fix/silk/fixed/encode_frame_FIX.c:263-423)."""
    _fields_ = [(k, C.c_int32) for k in ("maxBits", "useCBR", "condCoding", "nb_subfr", "frame_length", "started")] + [
        ("reserved0", C.c_int32 * 2), ("GainsUnq_Q16", C.c_int32 * 4), ("Gains_Q16", C.c_int32 * 4), ("lastGainIndexPrev", C.c_int32),
        ("LastGainIndex", C.c_int32), ("Lambda_Q10", C.c_int32), ("GainsIndices", C.c_int8 * 4)] + [(k, C.c_int32) for k in (
            "iter", "gainMult_Q8", "found_lower", "found_upper", "nBits_lower", "nBits_upper", "gainMult_lower", "gainMult_upper",
            "gainsID", "gainsID_lower", "gainsID_upper", "LastGainIndex_copy2", "done", "recode", "save2", "restore2", "nBits", "passes",
            "status", "reserved1")]


SIZES["silk_rate_ctl"] = C.sizeof(RateCtl)


class SilkStream(C.Structure):
    """opusgpu_silk_stream: what silk_encode_frame_FIX carries from one frame of a stream to the next (include/opusgpu_silk.h)."""
    _fields_ = [("x_buf", C.c_int16 * 400), ("prev_NLSFq_Q15", C.c_int16 * 16)] + [(k, C.c_int32) for k in (
        "prevLag", "prevSignalType", "first_frame_after_reset", "LTPCorr_Q15", "sum_log_gain_Q7", "LastGainIndex", "HarmBoost_smth_Q16",
        "HarmShapeGain_smth_Q16", "Tilt_smth_Q16", "ec_prevSignalType", "ec_prevLagIndex", "frameCounter")] + [("reserved", C.c_int32 * 4)]


SIZES["silk_stream"] = C.sizeof(SilkStream)


def silk_rate_control(rate_ctl, ec_state):
    """One step of silk_encode_frame_FIX's bitrate loop over a batch: rate_ctl uint8 [N][160] (opusgpu_silk_rate_ctl, updated in place),
    ec_state uint8 [N][1328] (the coder after the pass just coded; read only). Sets done / recode / save2 / restore2 per frame."""
    _check(rate_ctl, SIZES["silk_rate_ctl"], "rate_ctl")
    _check(ec_state, SIZES["ec_state"], "ec_state")
    n = rate_ctl.shape[0]
    if ec_state.shape[0] != n:
        raise ValueError("ec_state: %d records for %d frames" % (ec_state.shape[0], n))
    rc = _lib.load().opusgpu_silk_rate_control_batch(rate_ctl.data_ptr(), ec_state.data_ptr(), n, _lib.current_stream_handle())
    _lib.check(rc, "opusgpu_silk_rate_control_batch")
    return rate_ctl


class VadState(C.Structure):
    """opusgpu_vad_state == silk_VAD_state (opus-fix/silk/structs.h:60-73)."""
    _fields_ = [("AnaState", C.c_int32 * 2), ("AnaState1", C.c_int32 * 2), ("AnaState2", C.c_int32 * 2), ("XnrgSubfr", C.c_int32 * 4),
                ("NrgRatioSmth_Q8", C.c_int32 * 4), ("HPstate", C.c_int16), ("pad", C.c_int16), ("NL", C.c_int32 * 4), ("inv_NL", C.c_int32 * 4),
                ("NoiseLevelBias", C.c_int32 * 4), ("counter", C.c_int32)]


class VadIn(C.Structure):
    _fields_ = [("pIn", C.c_int16 * 320), ("frame_length", C.c_int32), ("fs_kHz", C.c_int32), ("reserved", C.c_int32 * 2)]


class VadOut(C.Structure):
    _fields_ = [("speech_activity_Q8", C.c_int32), ("input_tilt_Q15", C.c_int32), ("input_quality_bands_Q15", C.c_int32 * 4),
                ("status", C.c_int32), ("reserved", C.c_int32)]


SIZES["vad_in"] = C.sizeof(VadIn)
SIZES["vad_state"] = C.sizeof(VadState)
SIZES["vad_out"] = C.sizeof(VadOut)


def silk_VAD_GetSA_Q8(vad_in, vad_state, vad_out=None):
    """silk_VAD_GetSA_Q8_c() over a batch of records: vad_in uint8 [N][656], vad_state uint8 [N][112] (silk_VAD_state, updated in place)
    -> uint8 [N][32] (speech_activity_Q8, input_tilt_Q15, input_quality_bands_Q15[4], status)."""
    import torch
    _check(vad_in, SIZES["vad_in"], "vad_in")
    _check(vad_state, SIZES["vad_state"], "vad_state")
    n = vad_in.shape[0]
    if vad_state.shape[0] != n:
        raise ValueError("vad_state: %d records for %d inputs" % (vad_state.shape[0], n))
    if vad_out is None:
        vad_out = torch.empty((n, SIZES["vad_out"]), dtype=torch.uint8, device=vad_in.device)
    _check(vad_out, SIZES["vad_out"], "vad_out")
    rc = _lib.load().opusgpu_silk_vad_batch(vad_in.data_ptr(), vad_state.data_ptr(), vad_out.data_ptr(), n, _lib.current_stream_handle())
    _lib.check(rc, "opusgpu_silk_vad_batch")
    return vad_out

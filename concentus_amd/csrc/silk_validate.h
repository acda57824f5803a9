// silk_validate.h -- bounds checks of the device-resident SILK function-boundary records (include/opusgpu_silk.h).
//
// The batched SILK kernels index LDS, private arrays and the record's own arrays with header fields that arrive in
// device memory (nb_subfr, subfr_length, D, pitchL, ltp_mem_length ...): the host cannot look at them without a
// round trip, so every lane checks ITS record before touching anything, skips a record that would index out of bounds
// (outputs zeroed, NSQ state untouched, Burg res_nrg_Q = INT32_MIN) and counts it in the library's bad-record counter,
// which opusgpu_silk_bad_records() returns and clears. The limits are the reference's own (silk/define.h:
// MAX_NB_SUBFR 4, MAX_SUB_FRAME_LENGTH 80, MAX_FRAME_LENGTH 320, LTP_MEM_LENGTH_MS 20 -> 320, MAX_LPC_ORDER 16,
// MAX_SHAPE_LPC_ORDER 16, LTP_ORDER 5, MAX_DEL_DEC_STATES 4; burg_modified_FIX.c:36 MAX_FRAME_SIZE 384; the
// silk_assert( start_idx > 0 ) of NSQ.c:141 / NSQ_del_dec.c:236).
#pragma once
#include "../../include/opusgpu_silk.h"

namespace ca {

__device__ __forceinline__ bool burg_record_ok(const opusgpu_burg_in &in)
{
    const int L = in.subfr_length, n = in.nb_subfr, D = in.D;
    return n >= 1 && n <= 4 && D >= 1 && D <= OPUSGPU_SILK_MAX_ORDER && L > D && L <= OPUSGPU_SILK_BURG_MAX_X &&
           L * n <= OPUSGPU_SILK_BURG_MAX_X;
}

__device__ __forceinline__ bool find_lpc_record_ok(const opusgpu_find_lpc_in &in)
{
    const int L = in.subfr_length, n = in.nb_subfr, D = in.predictLPCOrder;
    if (!((n == 2 || n == 4) && (D == 10 || D == 16) && L >= D && L <= 80 && (L + D) * n <= OPUSGPU_SILK_BURG_MAX_X)) return false;
    for (int k = 0; k < D; k++)
        if (in.prev_NLSFq_Q15[k] < 0) return false;             // silk_NLSF2A indexes the cosine table with NLSF >> 8
    return true;
}

__device__ __forceinline__ bool process_nlsf_record_ok(const opusgpu_process_nlsf_in &in)
{
    const int D = in.predictLPCOrder;
    if (!((in.nb_subfr == 2 || in.nb_subfr == 4) && (D == 10 || D == 16))) return false;
    if ((unsigned)in.speech_activity_Q8 > 256u || (unsigned)in.useInterpolatedNLSFs > 1u || (unsigned)in.NLSFInterpCoef_Q2 > 4u) return false;
    if (in.NLSF_MSVQ_Survivors < 1 || in.NLSF_MSVQ_Survivors > 32 || (unsigned)in.signalType > 2u) return false;
    for (int k = 0; k < D; k++)
        if (in.NLSF_Q15[k] < 0 || in.prev_NLSFq_Q15[k] < 0) return false;
    return true;
}

__device__ __forceinline__ bool res_nrg_record_ok(const opusgpu_res_nrg_in &in)
{
    const int L = in.subfr_length, n = in.nb_subfr, D = in.LPC_order;
    if (!((n == 2 || n == 4) && D >= 2 && D <= 16 && !(D & 1) && L >= 1 && L <= 80 && (L + D) * n <= OPUSGPU_SILK_BURG_MAX_X)) return false;
    for (int k = 0; k < n; k++)
        if (in.gains[k] <= 0) return false;                      // quantisation gains are positive (silk_CLZ32(gain) - 1 is a shift count)
    return true;
}

__device__ __forceinline__ bool find_pred_coefs_record_ok(const opusgpu_find_pred_coefs_in &in)
{
    const int L = in.subfr_length, n = in.nb_subfr, D = in.predictLPCOrder, ltp = in.ltp_mem_length;
    if (!((n == 2 || n == 4) && (D == 10 || D == 16) && L >= D && L <= 80 && (L + D) * n <= OPUSGPU_SILK_BURG_MAX_X)) return false;
    if (!(ltp >= D && ltp <= OPUSGPU_SILK_MAX_LTP_MEM && n * L <= OPUSGPU_SILK_MAX_FRAME)) return false;
    if ((unsigned)in.signalType > 2u || (unsigned)in.useInterpolatedNLSFs > 1u || (unsigned)in.speech_activity_Q8 > 256u) return false;
    if (in.NLSF_MSVQ_Survivors < 1 || in.NLSF_MSVQ_Survivors > 32) return false;
    for (int k = 0; k < D; k++)
        if (in.prev_NLSFq_Q15[k] < 0) return false;
    for (int k = 0; k < n; k++) {
        if (in.Gains_Q16[k] <= 0) return false;
        // voiced: the lagged windows of find_LTP_FIX / LTP_analysis_filter_FIX must stay inside res_pitch[] / x[]
        if (in.signalType == 2 && !(in.pitchL[k] >= 2 && in.pitchL[k] + 2 + D <= ltp)) return false;
    }
    return true;
}

__device__ __forceinline__ bool process_gains_record_ok(const opusgpu_process_gains_in &in)
{
    if (!((in.nb_subfr == 2 || in.nb_subfr == 4) && in.subfr_length >= 1 && in.subfr_length <= 80)) return false;
    if ((unsigned)in.signalType > 2u || (unsigned)in.quantOffsetType > 1u || (unsigned)in.LastGainIndex > 63u) return false;
    for (int k = 0; k < in.nb_subfr; k++)
        if (in.Gains_Q16[k] <= 0 || in.ResNrg[k] < 0 || in.ResNrgQ[k] < -31 || in.ResNrgQ[k] > 31) return false;   // shift counts
    return true;
}

__device__ __forceinline__ bool noise_shape_record_ok(const opusgpu_noise_shape_in &in)
{
    const int n = in.nb_subfr, L = in.subfr_length, la = in.la_shape, W = in.shapeWinLength, D = in.shapingLPCOrder, fs = in.fs_kHz;
    if (!((n == 2 || n == 4) && (fs == 8 || fs == 12 || fs == 16) && L == 5 * fs && la >= 0 && la <= OPUSGPU_SILK_MAX_LA_SHAPE)) return false;
    if (!(W == L + 2 * la && W <= 240 && D >= 2 && D <= 16 && !(D & 1))) return false;
    // the two sine slopes: length a multiple of 4 in 16..120 (apply_sine_window_FIX.c:61-62)
    const int slope = (W - 3 * fs) >> 1;
    if (!(slope >= 16 && slope <= 120 && !(slope & 3) && 2 * slope + 3 * fs == W)) return false;
    if ((unsigned)in.signalType > 2u || (unsigned)in.speech_activity_Q8 > 256u || in.warping_Q16 < 0 || in.warping_Q16 > 32767) return false;
    if (in.signalType == 2)
        for (int k = 0; k < n; k++)
            if (in.pitchL[k] < 1) return false;                  // 3.0 / pitchL
    return true;
}

__device__ __forceinline__ bool prefilter_record_ok(const opusgpu_prefilter_in &in, const opusgpu_prefilter_state &st)
{
    const int n = in.nb_subfr, L = in.subfr_length, D = in.shapingLPCOrder;
    if (!((n == 2 || n == 4) && L >= 1 && L <= 80 && n * L <= OPUSGPU_SILK_MAX_FRAME && D >= 2 && D <= 16 && !(D & 1))) return false;
    if ((unsigned)in.signalType > 2u || in.warping_Q16 < 0 || in.warping_Q16 > 32767) return false;
    // the harmonic-shaping taps reach lag + 2 entries back: inside the part of the ring the kernel keeps (silk_shape_kernels.hip PF_RING = 320;
    // the encoder's lags end at 18 ms = 288 samples, pitch_est_defines.h PE_MAX_LAG_MS)
    if ((unsigned)st.sLTP_shp_buf_idx > 511u || st.lagPrev < 0 || st.lagPrev == 1 || st.lagPrev > 316) return false;
    for (int k = 0; k < n; k++)
        if (in.pitchL[k] < 0 || in.pitchL[k] == 1 || in.pitchL[k] > 316) return false;
    return true;
}

__device__ __forceinline__ bool find_pitch_lags_record_ok(const opusgpu_find_pitch_lags_in &in)
{
    const int fs = in.fs_kHz, n = in.nb_subfr, D = in.pitchEstimationLPCOrder, W = in.pitch_LPC_win_length, la = in.la_pitch;
    if (!((fs == 8 || fs == 16) && (n == 2 || n == 4) && in.frame_length == n * 5 * fs && in.ltp_mem_length == 20 * fs && la == 2 * fs)) return false;
    if (!(D >= 6 && D <= 16 && !(D & 1) && W >= 2 * la + D && W <= la + in.frame_length + in.ltp_mem_length && W <= 384)) return false;
    if ((unsigned)in.pitchEstimationComplexity > 2u || (unsigned)in.signalType > 2u || (unsigned)in.prevSignalType > 2u) return false;
    if ((unsigned)in.pitchEstimationThreshold_Q16 > 65536u || (unsigned)in.speech_activity_Q8 > 256u || in.prevLag < 0 || in.prevLag > 18 * fs) return false;
    return true;
}

__device__ __forceinline__ bool silk_bits_record_ok(const opusgpu_silk_bits_in &in, const opusgpu_ec_state &ec)
{
    const int fs = in.fs_kHz, n = in.nb_subfr, D = in.predictLPCOrder;
    if (!((fs == 8 || fs == 12 || fs == 16) && (n == 2 || n == 4) && (D == 10 || D == 16) && in.which >= 1 && in.which <= 3)) return false;
    if (in.frame_length < 16 || in.frame_length > OPUSGPU_SILK_MAX_FRAME || (in.frame_length & 15)) return false;
    if (ec.storage > OPUSGPU_EC_BUF || ec.offs > ec.storage || ec.end_offs > ec.storage || ec.rng <= 0x00800000u) return false;
    if ((unsigned)in.signalType > 2u || (unsigned)in.quantOffsetType > 1u || (unsigned)in.condCoding > 2u || (unsigned)in.Seed > 3u) return false;
    if (in.which & 1) {                                          // every symbol must lie inside its probability model
        const int cond = in.condCoding == 2;
        if (in.signalType == 0 && in.quantOffsetType > 1) return false;
        if (in.GainsIndices[0] < 0 || in.GainsIndices[0] >= (cond ? 41 : 64)) return false;
        for (int k = 1; k < n; k++)
            if (in.GainsIndices[k] < 0 || in.GainsIndices[k] >= 41) return false;
        if (in.NLSFIndices[0] < 0 || in.NLSFIndices[0] >= 32) return false;
        for (int k = 1; k <= D; k++)
            if (in.NLSFIndices[k] < -10 || in.NLSFIndices[k] > 10) return false;
        if ((unsigned)in.NLSFInterpCoef_Q2 > 4u) return false;
        if (in.signalType == 2) {
            const int ncont = n == 4 ? (fs == 8 ? 11 : 34) : (fs == 8 ? 3 : 12);
            if (in.lagIndex < 0 || in.lagIndex >= 16 * fs || in.contourIndex < 0 || in.contourIndex >= ncont) return false;
            if ((unsigned)in.PERIndex > 2u || (unsigned)in.LTP_scaleIndex > 2u) return false;
            for (int k = 0; k < n; k++)
                if (in.LTPIndex[k] < 0 || in.LTPIndex[k] >= (8 << in.PERIndex)) return false;
        }
    }
    return true;
}

__device__ __forceinline__ bool vad_record_ok(const opusgpu_vad_in &in, const opusgpu_vad_state &st)
{
    const int fl = in.frame_length, fs = in.fs_kHz;
    if (!((fs == 8 || fs == 12 || fs == 16) && (fl == 10 * fs || fl == 20 * fs) && fl <= OPUSGPU_SILK_MAX_FRAME && !(fl & 7))) return false;
    if (st.counter < 0) return false;
    for (int k = 0; k < 4; k++)
        if (st.NL[k] < 0 || st.inv_NL[k] <= 0 || st.NoiseLevelBias[k] < 1 || st.XnrgSubfr[k] < 0) return false;     // divisors / saturating sums
    return true;
}

__device__ __forceinline__ bool rate_ctl_record_ok(const opusgpu_silk_rate_ctl &c, unsigned rng)
{
    if (!((c.nb_subfr == 2 || c.nb_subfr == 4) && c.frame_length >= 80 && c.frame_length <= OPUSGPU_SILK_MAX_FRAME)) return false;   // a divisor
    if ((unsigned)c.condCoding > 2u || (unsigned)c.useCBR > 1u || c.maxBits < 0 || rng == 0) return false;
    if (c.LastGainIndex < -128 || c.LastGainIndex > 127 || c.lastGainIndexPrev < -128 || c.lastGainIndexPrev > 127) return false;    // opus_int8 fields
    if (c.started && ((unsigned)c.iter > 6u || c.gainMult_Q8 < -32768 || c.gainMult_Q8 > 32767)) return false;
    return true;
}

__device__ __forceinline__ bool nsq_record_ok(const opusgpu_nsq_in &in, int lagPrev)
{
    const int n = in.nb_subfr, L = in.subfr_length, ltp = in.ltp_mem_length, po = in.predictLPCOrder, so = in.shapingLPCOrder;
    if (!(n >= 1 && n <= 4 && L >= 1 && L <= 80 && in.frame_length == n * L && in.frame_length <= OPUSGPU_SILK_MAX_FRAME)) return false;
    if (!(ltp >= 1 && ltp <= OPUSGPU_SILK_MAX_FRAME && po >= 2 && po <= 16 && so >= 2 && so <= 16 && !(so & 1))) return false;
    if (in.signalType < 0 || in.signalType > 2 || (unsigned)in.quantOffsetType > 1u) return false;
    if (lagPrev < 0 || lagPrev > ltp) return false;
    if (in.signalType == 2)
        for (int k = 0; k < n; k++) {
            const int lag = in.pitchL[k];
            // re-whitening starts at ltp_mem_length - lag - predictLPCOrder - LTP_ORDER/2 (> 0 in the reference); the LTP taps
            // reach lag + LTP_ORDER/2 samples back
            if (lag < 3 || ltp - lag - po - 2 <= 0) return false;
        }
    return true;
}

__device__ __forceinline__ bool nsq_dd_record_ok(const opusgpu_nsq_dd_in &in, int lagPrev)
{
    if (!nsq_record_ok(in.base, lagPrev)) return false;
    if (in.nStatesDelayedDecision < 1 || in.nStatesDelayedDecision > OPUSGPU_SILK_MAX_DEL_DEC_STATES) return false;
    // decisionDelay = min(DECISION_DELAY, subfr_length, lag - LTP_ORDER/2 - 1) must stay positive (NSQ_del_dec.c:157-170)
    if (in.base.signalType == 2) {
        for (int k = 0; k < in.base.nb_subfr; k++)
            if (in.base.pitchL[k] - 3 < 1) return false;
    } else if (lagPrev > 0 && lagPrev - 3 < 1) {
        return false;
    }
    return true;
}

}  // namespace ca

// silk_pitch_dev.h -- silk_find_pitch_lags_FIX (opus-fix/silk/fixed/find_pitch_lags_FIX.c:37-145) and the three-stage pitch
// estimator it calls (seventh slice): LPC whitening of the pitch-analysis buffer (the `res_pitch` signal that
// silk_noise_shape_analysis_FIX and silk_find_pred_coefs_FIX consume), voicing decision, pitch lags, lag / contour indices.
//
//   silk_find_pitch_lags_FIX                  opus-fix/silk/fixed/find_pitch_lags_FIX.c:37-145
//   silk_pitch_analysis_core                  opus-fix/silk/fixed/pitch_analysis_core_FIX.c:86-581
//   silk_P_Ana_calc_corr_st3 / _energy_st3    opus-fix/silk/fixed/pitch_analysis_core_FIX.c:598-746
//   silk_resampler_down2                      opus-fix/silk/resampler_down2.c:36-74
//   silk_insertion_sort_decreasing_int16      opus-fix/silk/sort.c:88-130
//   silk_schur / silk_k2a / silk_bwexpander   opus-fix/silk/fixed/schur_FIX.c:36-106, k2a_FIX.c:35-53, opus-fix/silk/bwexpander.c:35-51
//   silk_LPC_analysis_filter                  opus-fix/silk/LPC_analysis_filter.c:41-108 (FIXED_POINT branch: celt_fir)
//   celt_pitch_xcorr (as called here)         opus-fix/celt/pitch.c:251-285: xcorr[i] = sum_j x[j] * y[j + i], MAC16_16 sums that wrap
//   lag code books                            opus-fix/silk/pitch_est_tables.c:34-99
//
// One lane owns one frame. 8 and 16 kHz inputs (the 12 kHz path needs silk_resampler_down2_3 and is reported as unsupported
// by the kernel's record check).
#pragma once
#include "silk_shape_dev.h"

namespace ca {

enum { PE_MAX_NB_SUBFR = 4, PE_SUBFR_LENGTH_MS = 5, PE_LTP_MEM_LENGTH_MS = 20, PE_MAX_LAG_MS = 18, PE_MIN_LAG_MS = 2, PE_D_SRCH_LENGTH = 24,
       PE_NB_STAGE3_LAGS = 5, PE_NB_CBKS_STAGE2 = 3, PE_NB_CBKS_STAGE2_EXT = 11, PE_NB_CBKS_STAGE3_MAX = 34, PE_NB_CBKS_STAGE3_10MS = 12,
       PE_NB_CBKS_STAGE2_10MS = 3,
       SF_LENGTH_4KHZ = PE_SUBFR_LENGTH_MS * 4, SF_LENGTH_8KHZ = PE_SUBFR_LENGTH_MS * 8, MIN_LAG_4KHZ = PE_MIN_LAG_MS * 4,
       MIN_LAG_8KHZ = PE_MIN_LAG_MS * 8, MAX_LAG_4KHZ = PE_MAX_LAG_MS * 4, MAX_LAG_8KHZ = PE_MAX_LAG_MS * 8 - 1,
       CSTRIDE_4KHZ = MAX_LAG_4KHZ + 1 - MIN_LAG_4KHZ, CSTRIDE_8KHZ = MAX_LAG_8KHZ + 3 - (MIN_LAG_8KHZ - 2), D_COMP_MIN = MIN_LAG_8KHZ - 3,
       D_COMP_MAX = MAX_LAG_8KHZ + 4, D_COMP_STRIDE = D_COMP_MAX - D_COMP_MIN, PE_SCRATCH_SIZE = 22,
       PE_MAX_FRAME_8KHZ = (PE_LTP_MEM_LENGTH_MS + PE_MAX_NB_SUBFR * PE_SUBFR_LENGTH_MS) * 8, PE_MAX_FRAME = PE_MAX_FRAME_8KHZ * 2 };

// pitch_est_tables.c:34-99 (small hand-made tables of the specification, RFC 6716 section 4.2.7.6.1; tests/test_tables.py
// compares them with the compiled reference)
CA_DEVICE_CONST i8 SILK_CB_lags_stage2_10_ms[2 * 3] = {0, 1, 0, 0, 0, 1};
CA_DEVICE_CONST i8 SILK_CB_lags_stage3_10_ms[2 * 12] = {0, 0, 1, -1, 1, -1, 2, -2, 2, -2, 3, -3, 0, 1, 0, 1, -1, 2, -1, 2, -2, 3, -2, 3};
CA_DEVICE_CONST i8 SILK_Lag_range_stage3_10_ms[2 * 2] = {-3, 7, -2, 7};
CA_DEVICE_CONST i8 SILK_CB_lags_stage2[4 * 11] = {
    0, 2, -1, -1, -1, 0, 0, 1, 1, 0, 1,
    0, 1, 0, 0, 0, 0, 0, 1, 0, 0, 0,
    0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 0,
    0, -1, 2, 1, 0, 1, 1, 0, 0, -1, -1,
};
CA_DEVICE_CONST i8 SILK_CB_lags_stage3[4 * 34] = {
    0, 0, 1, -1, 0, 1, -1, 0, -1, 1, -2, 2, -2, -2, 2, -3, 2, 3, -3, -4, 3, -4, 4, 4, -5, 5, -6, -5, 6, -7, 6, 5, 8, -9,
    0, 0, 1, 0, 0, 0, 0, 0, 0, 0, -1, 1, 0, 0, 1, -1, 0, 1, -1, -1, 1, -1, 2, 1, -1, 2, -2, -2, 2, -2, 2, 2, 3, -3,
    0, 1, 0, 0, 0, 0, 0, 0, 1, 0, 1, 0, 0, 1, -1, 1, 0, 0, 2, 1, -1, 2, -1, -1, 2, -1, 2, 2, -1, 3, -2, -2, -2, 3,
    0, 1, 0, 0, 1, 0, 1, -1, 2, -1, 2, -1, 2, 3, -2, 3, -2, -2, 4, 4, -3, 5, -3, -4, 6, -4, 6, 5, -5, 8, -6, -5, -7, 9,
};
CA_DEVICE_CONST i8 SILK_Lag_range_stage3[3 * 4 * 2] = {
    -5, 8, -1, 6, -1, 6, -4, 10,
    -6, 10, -2, 6, -1, 6, -5, 10,
    -9, 12, -3, 7, -2, 7, -7, 13,
};
CA_DEVICE_CONST i8 SILK_nb_cbk_searchs_stage3[3] = {16, 24, 34};

template <class XA>
CA_DEV void silk_resampler_down2_dev(i32 *S, i16 *out, XA in, int inLen)                    // resampler_down2.c:36-74
{
    const int len2 = inLen >> 1;
#pragma unroll 4
    for (int k = 0; k < len2; k++) {
        i32 in32 = shl32((i32)in[2 * k], 10);
        i32 Y = s_subw(in32, S[0]);
        i32 X = s_smlawb(Y, Y, 39809 - 65536);                                              // silk_resampler_down2_1
        i32 out32 = s_addw(S[0], X);
        S[0] = s_addw(in32, X);
        in32 = shl32((i32)in[2 * k + 1], 10);
        Y = s_subw(in32, S[1]);
        X = s_smulwb(Y, 9872);                                                              // silk_resampler_down2_0
        out32 = s_addw(out32, S[1]);
        out32 = s_addw(out32, X);
        S[1] = s_addw(in32, X);
        const i32 v = s_rshift_round(out32, 11);
        out[k] = (i16)(v > 32767 ? 32767 : (v < -32768 ? -32768 : v));
    }
}

CA_DEV void silk_insertion_sort_decreasing_int16_dev(i16 *a, int *idx, int L, int K)        // sort.c:88-130
{
    for (int i = 0; i < K; i++) idx[i] = i;
    for (int i = 1; i < K; i++) {
        const int value = a[i];
        int j;
        for (j = i - 1; j >= 0 && value > a[j]; j--) { a[j + 1] = a[j]; idx[j + 1] = idx[j]; }
        a[j + 1] = (i16)value;
        idx[j + 1] = i;
    }
    for (int i = K; i < L; i++) {
        const int value = a[i];
        if (value > a[K - 1]) {
            int j;
            for (j = K - 2; j >= 0 && value > a[j]; j--) { a[j + 1] = a[j]; idx[j + 1] = idx[j]; }
            a[j + 1] = (i16)value;
            idx[j + 1] = i;
        }
    }
}

template <class XA, class YA>
CA_DEV i32 pe_inner_prod(XA x, YA y, int len)                                               // silk_inner_prod_aligned = celt_inner_prod
{
    i32 s = 0;
#pragma unroll 8
    for (int i = 0; i < len; i++) s = s_addw(s, __mul24((i32)x[i], (i32)y[i]));
    return s;
}

// Correlations of a 40-sample target against a basis that slides DOWN one sample per lag: sink(j, <tgt, sig[b0 - j ..)>, entering,
// leaving) for j = 0 .. nlags - 1, where entering = sig[b0 - j] and leaving = sig[b0 - j + 40] (the samples the basis gains / loses
// when it moves from lag j - 1 to lag j; for j = 0: its first sample and the one after its last). Target and basis live in
// registers; every signal sample is read once instead of once per lag. `acc0`: per-lag values to add to (for targets longer than
// 40: call again with the next 40 samples), or nullptr.
enum { PE_BLK = 40 };
template <class TA, class XA, class F>
CA_DEV void pe_xcorr_sliding(TA tgt, XA sig, int b0, int nlags, F sink)
{
    i32 t[PE_BLK], b[PE_BLK];
#pragma unroll
    for (int i = 0; i < PE_BLK; i++) { t[i] = (i32)tgt[i]; b[i] = (i32)sig[b0 + i]; }
    i32 leaving = (i32)sig[b0 + PE_BLK];
#pragma unroll 2
    for (int j = 0;; j++) {
        i32 cc = 0;
#pragma unroll
        for (int i = 0; i < PE_BLK; i++) cc = s_addw(cc, __mul24(t[i], b[i]));
        sink(j, cc, b[0], leaving);
        if (j + 1 >= nlags) break;
        leaving = b[PE_BLK - 1];
#pragma unroll
        for (int i = PE_BLK - 1; i > 0; i--) b[i] = b[i - 1];
        b[0] = (i32)sig[b0 - j - 1];
    }
}

// pitch_analysis_core_FIX.c:598-746: correlations and energies of every subframe against the lags the stage-3 code books can
// reach; `frame` is the (possibly down-shifted) full-rate signal. The reference expands both into [subframe][code book vector][5]
// arrays (2 x 2.7 KB); kept here as the per-subframe lag tables they are copied from (2 x 4 x 22 values) -- the search below
// indexes them with the same `Lag_CB - lag_low + lag_counter` the copy loop used.
template <class XA>
CA_DEV void pe_calc_corr_energy_st3(i32 *corr /*[4][PE_SCRATCH_SIZE]*/, i32 *nrg /*[4][PE_SCRATCH_SIZE]*/, XA frame, int start_lag, int sf_length,
                                    int nb_subfr, const i8 *Lag_range_ptr)
{
    for (int k = 0; k < nb_subfr; k++) {
        const int t0 = 4 * sf_length + k * sf_length;                                       // target_ptr
        const int lag_low = Lag_range_ptr[k * 2], lag_high = Lag_range_ptr[k * 2 + 1];
        i32 *scratch_c = corr + k * PE_SCRATCH_SIZE, *scratch_e = nrg + k * PE_SCRATCH_SIZE;
        // correlations: scratch[j - lag_low] = <target, target - start_lag - j>, j = lag_low .. lag_high: the basis slides down with j
        for (int j = 0; j <= lag_high - lag_low; j++) scratch_c[j] = 0;
        for (int h = 0; h < sf_length; h += PE_BLK)
            pe_xcorr_sliding(frame + (t0 + h), frame, t0 + h - start_lag - lag_low, lag_high - lag_low + 1,
                             [&](int j, i32 cc, i32, i32) { scratch_c[j] = s_addw(scratch_c[j], cc); });
        // energies, recursively from the first lag (:713-727)
        const int b0 = t0 - (start_lag + lag_low);                                          // basis_ptr
        i32 energy = pe_inner_prod(frame + b0, frame + b0, sf_length);
        scratch_e[0] = energy;
        const int lag_diff = lag_high - lag_low + 1;
#pragma unroll 4
        for (int i = 1; i < lag_diff; i++) {
            const i32 a = frame[b0 + sf_length - i], b = frame[b0 - i];
            energy -= __mul24(a, a);
            energy = s_add_sat32(energy, __mul24(b, b));
            scratch_e[i] = energy;
        }
    }
}

// pitch_analysis_core_FIX.c:86-581; returns 0 voiced / 1 unvoiced. `scr`: frame_length samples of scratch for the down-shifted
// copy of the input that stage 3 may need.
template <class XA, class SCR>
CA_DEV int silk_pitch_analysis_core_geom_dev(XA frame, SCR scr, int *pitch_out, int *lagIndex, int *contourIndex, int *LTPCorr_Q15, int prevLag,
                                             i32 search_thres1_Q16, int search_thres2_Q13, const int Fs_kHz, int complexity, const int nb_subfr)
{
    const int frame_length = (PE_LTP_MEM_LENGTH_MS + nb_subfr * PE_SUBFR_LENGTH_MS) * Fs_kHz;
    const int frame_length_4kHz = (PE_LTP_MEM_LENGTH_MS + nb_subfr * PE_SUBFR_LENGTH_MS) * 4;
    const int frame_length_8kHz = (PE_LTP_MEM_LENGTH_MS + nb_subfr * PE_SUBFR_LENGTH_MS) * 8;
    const int sf_length = PE_SUBFR_LENGTH_MS * Fs_kHz, min_lag = PE_MIN_LAG_MS * Fs_kHz, max_lag = PE_MAX_LAG_MS * Fs_kHz - 1;
    i16 frame_8kHz[PE_MAX_FRAME_8KHZ], frame_4kHz[PE_MAX_FRAME_8KHZ / 2];
    i32 filt_state[2];
    if (Fs_kHz == 16) {
        filt_state[0] = filt_state[1] = 0;
        silk_resampler_down2_dev(filt_state, frame_8kHz, frame, frame_length);
    } else {
        for (int i = 0; i < frame_length_8kHz; i++) frame_8kHz[i] = (i16)(i32)frame[i];
    }
    filt_state[0] = filt_state[1] = 0;
    silk_resampler_down2_dev(filt_state, frame_4kHz, (const i16 *)frame_8kHz, frame_length_8kHz);
    {
        i32 hi = frame_4kHz[frame_length_4kHz - 1];                                        // every sample is read once, on its way down
#pragma unroll 4
        for (int i = frame_length_4kHz - 1; i > 0; i--) {
            const i32 lo = frame_4kHz[i - 1], v = hi + lo;
            frame_4kHz[i] = (i16)(v > 32767 ? 32767 : (v < -32768 ? -32768 : v));
            hi = lo;
        }
    }
    i32 energy;
    int shift;
    silk_sum_sqr_shift_dev(&energy, &shift, (const i16 *)frame_4kHz, frame_length_4kHz);
    if (shift > 0) {
        shift >>= 1;
#pragma unroll 8
        for (int i = 0; i < frame_length_4kHz; i++) frame_4kHz[i] = (i16)(frame_4kHz[i] >> shift);
    }
    // ---- first stage, 4 kHz (:170-289)
    i16 C[PE_MAX_NB_SUBFR * CSTRIDE_8KHZ];
    for (int i = 0; i < (nb_subfr >> 1) * CSTRIDE_4KHZ; i++) C[i] = 0;
    for (int k = 0; k < (nb_subfr >> 1); k++) {
        const int t0 = 4 * SF_LENGTH_4KHZ + k * SF_LENGTH_8KHZ;
        const i16 *target_ptr = &frame_4kHz[t0];
        static_assert(SF_LENGTH_8KHZ == PE_BLK, "one block per target");
        i32 normalizer = pe_inner_prod(target_ptr, target_ptr, SF_LENGTH_8KHZ);
        normalizer = s_addw(normalizer, pe_inner_prod(target_ptr - MIN_LAG_4KHZ, target_ptr - MIN_LAG_4KHZ, SF_LENGTH_8KHZ));
        normalizer = s_addw(normalizer, s_smulbb(SF_LENGTH_8KHZ, 4000));
        // lag d = MIN_LAG_4KHZ + j: xcorr32[MAX_LAG_4KHZ - d] = <target, target - d>; the normaliser follows the basis' energy
        pe_xcorr_sliding(target_ptr, (const i16 *)frame_4kHz, t0 - MIN_LAG_4KHZ, MAX_LAG_4KHZ - MIN_LAG_4KHZ + 1,
                         [&](int j, i32 cross_corr, i32 entering, i32 leaving) {
                             if (j > 0) normalizer = s_addw(normalizer, __mul24(entering, entering) - __mul24(leaving, leaving));
                             C[k * CSTRIDE_4KHZ + j] = (i16)s_div32_varq(cross_corr, normalizer, 13 + 1);
                         });
    }
    if (nb_subfr == PE_MAX_NB_SUBFR) {
        for (int i = MAX_LAG_4KHZ; i >= MIN_LAG_4KHZ; i--) {
            i32 sum = (i32)C[i - MIN_LAG_4KHZ] + (i32)C[CSTRIDE_4KHZ + i - MIN_LAG_4KHZ];
            sum = s_smlawb(sum, sum, shl32(-i, 4));
            C[i - MIN_LAG_4KHZ] = (i16)sum;
        }
    } else {
        for (int i = MAX_LAG_4KHZ; i >= MIN_LAG_4KHZ; i--) {
            i32 sum = shl32((i32)C[i - MIN_LAG_4KHZ], 1);
            sum = s_smlawb(sum, sum, shl32(-i, 4));
            C[i - MIN_LAG_4KHZ] = (i16)sum;
        }
    }
    int d_srch[PE_D_SRCH_LENGTH];
    int length_d_srch = 4 + (complexity << 1);
    silk_insertion_sort_decreasing_int16_dev(C, d_srch, CSTRIDE_4KHZ, length_d_srch);
    const int Cmax = C[0];
    if (Cmax < 3277) {                                                                      // SILK_FIX_CONST(0.2, 14)
        for (int k = 0; k < nb_subfr; k++) pitch_out[k] = 0;
        *LTPCorr_Q15 = 0; *lagIndex = 0; *contourIndex = 0;
        return 1;
    }
    const i32 threshold = s_smulwb(search_thres1_Q16, Cmax);
    for (int i = 0; i < length_d_srch; i++) {
        if (C[i] > threshold) d_srch[i] = shl32(d_srch[i] + MIN_LAG_4KHZ, 1);
        else { length_d_srch = i; break; }
    }
    i16 d_comp[D_COMP_STRIDE];
    for (int i = D_COMP_MIN; i < D_COMP_MAX; i++) d_comp[i - D_COMP_MIN] = 0;
    for (int i = 0; i < length_d_srch; i++) d_comp[d_srch[i] - D_COMP_MIN] = 1;
    for (int i = D_COMP_MAX - 1; i >= MIN_LAG_8KHZ; i--) d_comp[i - D_COMP_MIN] += d_comp[i - 1 - D_COMP_MIN] + d_comp[i - 2 - D_COMP_MIN];
    length_d_srch = 0;
    for (int i = MIN_LAG_8KHZ; i < MAX_LAG_8KHZ + 1; i++) {
        if (d_comp[i + 1 - D_COMP_MIN] > 0) { d_srch[length_d_srch] = i; length_d_srch++; }
    }
    for (int i = D_COMP_MAX - 1; i >= MIN_LAG_8KHZ; i--)
        d_comp[i - D_COMP_MIN] += d_comp[i - 1 - D_COMP_MIN] + d_comp[i - 2 - D_COMP_MIN] + d_comp[i - 3 - D_COMP_MIN];
    int length_d_comp = 0;
    for (int i = MIN_LAG_8KHZ; i < D_COMP_MAX; i++) {
        if (d_comp[i - D_COMP_MIN] > 0) { d_comp[length_d_comp] = (i16)(i - 2); length_d_comp++; }
    }
    // ---- second stage, 8 kHz (:291-434)
    silk_sum_sqr_shift_dev(&energy, &shift, (const i16 *)frame_8kHz, frame_length_8kHz);
    if (shift > 0) {
        shift >>= 1;
#pragma unroll 8
        for (int i = 0; i < frame_length_8kHz; i++) frame_8kHz[i] = (i16)(frame_8kHz[i] >> shift);
    }
#pragma unroll 8
    for (int i = 0; i < nb_subfr * CSTRIDE_8KHZ; i++) C[i] = 0;
    for (int k = 0; k < nb_subfr; k++) {
        const i16 *target_ptr = &frame_8kHz[PE_LTP_MEM_LENGTH_MS * 8 + k * SF_LENGTH_8KHZ];
        i32 t[SF_LENGTH_8KHZ];                                                             // the target in registers: read once per subframe
        i32 energy_target = 1;
#pragma unroll
        for (int i = 0; i < SF_LENGTH_8KHZ; i++) { t[i] = (i32)target_ptr[i]; energy_target = s_addw(energy_target, __mul24(t[i], t[i])); }
        for (int j = 0; j < length_d_comp; j++) {
            const int d = d_comp[j];
            const i16 *basis_ptr = target_ptr - d;
            i32 b[SF_LENGTH_8KHZ], cross_corr = 0;                                         // the basis too: one read serves both sums
#pragma unroll
            for (int i = 0; i < SF_LENGTH_8KHZ; i++) { b[i] = (i32)basis_ptr[i]; cross_corr = s_addw(cross_corr, __mul24(t[i], b[i])); }
            if (cross_corr > 0) {
                i32 energy_basis = 0;
#pragma unroll
                for (int i = 0; i < SF_LENGTH_8KHZ; i++) energy_basis = s_addw(energy_basis, __mul24(b[i], b[i]));
                C[k * CSTRIDE_8KHZ + d - (MIN_LAG_8KHZ - 2)] = (i16)s_div32_varq(cross_corr, s_addw(energy_target, energy_basis), 13 + 1);
            } else {
                C[k * CSTRIDE_8KHZ + d - (MIN_LAG_8KHZ - 2)] = 0;
            }
        }
    }
    i32 CCmax = (i32)0x80000000, CCmax_b = (i32)0x80000000;
    int CBimax = 0, lag = -1;
    i32 prevLag_log2_Q7 = 0;
    if (prevLag > 0) {
        if (Fs_kHz == 12) prevLag = shl32(prevLag, 1) / 3;
        else if (Fs_kHz == 16) prevLag >>= 1;
        prevLag_log2_Q7 = s_lin2log((i32)prevLag);
    }
    int cbk_size, nb_cbk_search;
    const i8 *Lag_CB_ptr;
    if (nb_subfr == PE_MAX_NB_SUBFR) {
        cbk_size = PE_NB_CBKS_STAGE2_EXT;
        Lag_CB_ptr = SILK_CB_lags_stage2;
        nb_cbk_search = (Fs_kHz == 8 && complexity > 0) ? PE_NB_CBKS_STAGE2_EXT : PE_NB_CBKS_STAGE2;
    } else {
        cbk_size = PE_NB_CBKS_STAGE2_10MS;
        Lag_CB_ptr = SILK_CB_lags_stage2_10_ms;
        nb_cbk_search = PE_NB_CBKS_STAGE2_10MS;
    }
    for (int k = 0; k < length_d_srch; k++) {
        const int d = d_srch[k];
        i32 CCmax_new = (i32)0x80000000;
        int CBimax_new = 0;
        for (int j = 0; j < nb_cbk_search; j++) {
            i32 cc = 0;
            for (int i = 0; i < nb_subfr; i++) cc += (i32)C[i * CSTRIDE_8KHZ + d + Lag_CB_ptr[i * cbk_size + j] - (MIN_LAG_8KHZ - 2)];
            if (cc > CCmax_new) { CCmax_new = cc; CBimax_new = j; }
        }
        const i32 lag_log2_Q7 = s_lin2log(d);
        i32 CCmax_new_b = CCmax_new - (s_smulbb(nb_subfr * 1638, lag_log2_Q7) >> 7);       // PE_SHORTLAG_BIAS Q13
        if (prevLag > 0) {
            i32 delta_lag_log2_sqr_Q7 = lag_log2_Q7 - prevLag_log2_Q7;
            delta_lag_log2_sqr_Q7 = s_smulbb(delta_lag_log2_sqr_Q7, delta_lag_log2_sqr_Q7) >> 7;
            i32 prev_lag_bias_Q13 = s_smulbb(nb_subfr * 1638, *LTPCorr_Q15) >> 15;         // PE_PREVLAG_BIAS Q13
            prev_lag_bias_Q13 = s_mulw(prev_lag_bias_Q13, delta_lag_log2_sqr_Q7) / (delta_lag_log2_sqr_Q7 + 64);
            CCmax_new_b -= prev_lag_bias_Q13;
        }
        if (CCmax_new_b > CCmax_b && CCmax_new > s_smulbb(nb_subfr, search_thres2_Q13) && SILK_CB_lags_stage2[CBimax_new] <= MIN_LAG_8KHZ) {
            CCmax_b = CCmax_new_b;
            CCmax = CCmax_new;
            lag = d;
            CBimax = CBimax_new;
        }
    }
    if (lag == -1) {
        for (int k = 0; k < nb_subfr; k++) pitch_out[k] = 0;
        *LTPCorr_Q15 = 0; *lagIndex = 0; *contourIndex = 0;
        return 1;
    }
    *LTPCorr_Q15 = shl32(CCmax / nb_subfr, 2);
    if (Fs_kHz > 8) {
        // ---- third stage, input rate (:447-563)
        silk_sum_sqr_shift_dev(&energy, &shift, frame, frame_length);
        const bool shifted = shift > 0;
        if (shifted) {
            shift >>= 1;
#pragma unroll 8
            for (int i = 0; i < frame_length; i++) scr[i] = (i16)((i32)frame[i] >> shift);
        } else if ((const void *)&scr[0] != (const void *)&frame[0]) {                      // (the caller may hand over the frame in `scr` itself)
            for (int i = 0; i < frame_length; i++) scr[i] = (i16)(i32)frame[i];
        }
        const int CBimax_old = CBimax;
        if (Fs_kHz == 12) lag = s_smulbb(lag, 3) >> 1;
        else if (Fs_kHz == 16) lag = shl32(lag, 1);
        else lag = s_smulbb(lag, 3);
        lag = s_limit(lag, min_lag, max_lag);
        const int start_lag = imax(lag - 2, min_lag), end_lag = imin(lag + 2, max_lag);
        int lag_new = lag;
        CBimax = 0;
        CCmax = (i32)0x80000000;
        for (int k = 0; k < nb_subfr; k++) pitch_out[k] = lag + 2 * SILK_CB_lags_stage2[k * PE_NB_CBKS_STAGE2_EXT + CBimax_old];
        const i8 *Lag_range_ptr;
        if (nb_subfr == PE_MAX_NB_SUBFR) {
            nb_cbk_search = SILK_nb_cbk_searchs_stage3[complexity];
            cbk_size = PE_NB_CBKS_STAGE3_MAX;
            Lag_CB_ptr = SILK_CB_lags_stage3;
            Lag_range_ptr = &SILK_Lag_range_stage3[complexity * 8];
        } else {
            nb_cbk_search = PE_NB_CBKS_STAGE3_10MS;
            cbk_size = PE_NB_CBKS_STAGE3_10MS;
            Lag_CB_ptr = SILK_CB_lags_stage3_10_ms;
            Lag_range_ptr = SILK_Lag_range_stage3_10_ms;
        }
        i32 corr_st3[PE_MAX_NB_SUBFR * PE_SCRATCH_SIZE], nrg_st3[PE_MAX_NB_SUBFR * PE_SCRATCH_SIZE];
        pe_calc_corr_energy_st3(corr_st3, nrg_st3, scr, start_lag, sf_length, nb_subfr, Lag_range_ptr);
        int lag_counter = 0;
        const i32 contour_bias_Q15 = 1638 / lag;                                           // PE_FLATCONTOUR_BIAS Q15
        const i32 energy_target = s_addw(pe_inner_prod(scr + PE_LTP_MEM_LENGTH_MS * Fs_kHz, scr + PE_LTP_MEM_LENGTH_MS * Fs_kHz, nb_subfr * sf_length), 1);
        for (int d = start_lag; d <= end_lag; d++) {
            for (int j = 0; j < nb_cbk_search; j++) {
                i32 cross_corr = 0;
                energy = energy_target;
                for (int k = 0; k < nb_subfr; k++) {
                    const int idx = k * PE_SCRATCH_SIZE + Lag_CB_ptr[k * cbk_size + j] - Lag_range_ptr[k * 2] + lag_counter;
                    cross_corr = s_addw(cross_corr, corr_st3[idx]);
                    energy = s_addw(energy, nrg_st3[idx]);
                }
                i32 CCmax_new = 0;
                if (cross_corr > 0) {
                    CCmax_new = s_div32_varq(cross_corr, energy, 13 + 1);
                    const i32 diff = 32767 - s_mulw(contour_bias_Q15, j);
                    CCmax_new = s_smulwb(CCmax_new, diff);
                }
                if (CCmax_new > CCmax && (d + SILK_CB_lags_stage3[j]) <= max_lag) {
                    CCmax = CCmax_new;
                    lag_new = d;
                    CBimax = j;
                }
            }
            lag_counter++;
        }
        for (int k = 0; k < nb_subfr; k++) {
            pitch_out[k] = lag_new + Lag_CB_ptr[k * cbk_size + CBimax];
            pitch_out[k] = s_limit(pitch_out[k], min_lag, PE_MAX_LAG_MS * Fs_kHz);
        }
        *lagIndex = (i16)(lag_new - min_lag);
        *contourIndex = (i8)CBimax;
    } else {
        for (int k = 0; k < nb_subfr; k++) {
            pitch_out[k] = lag + Lag_CB_ptr[k * cbk_size + CBimax];
            pitch_out[k] = s_limit(pitch_out[k], MIN_LAG_8KHZ, PE_MAX_LAG_MS * 8);
        }
        *lagIndex = (i16)(lag - MIN_LAG_8KHZ);
        *contourIndex = (i8)CBimax;
    }
    return 0;
}

CA_DEV i32 silk_schur_dev(i16 *rc_Q15, const i32 *c, int order)                             // schur_FIX.c:36-106
{
    i32 C0[SILK_MAX_LPC + 1], C1[SILK_MAX_LPC + 1];
    int lz = s_clz32(c[0]);
    if (lz < 2) {
        for (int k = 0; k < order + 1; k++) C0[k] = C1[k] = c[k] >> 1;
    } else if (lz > 2) {
        lz -= 2;
        for (int k = 0; k < order + 1; k++) C0[k] = C1[k] = shl32(c[k], lz);
    } else {
        for (int k = 0; k < order + 1; k++) C0[k] = C1[k] = c[k];
    }
    int k;
    for (k = 0; k < order; k++) {
        if (s_abs(C0[k + 1]) >= C1[0]) {
            rc_Q15[k] = (i16)(C0[k + 1] > 0 ? -32440 : 32440);                              // SILK_FIX_CONST(.99f, 15)
            k++;
            break;
        }
        i32 rc_tmp_Q15 = -(C0[k + 1] / imax(C1[0] >> 15, 1));
        rc_tmp_Q15 = rc_tmp_Q15 > 32767 ? 32767 : (rc_tmp_Q15 < -32768 ? -32768 : rc_tmp_Q15);
        rc_Q15[k] = (i16)rc_tmp_Q15;
        for (int n = 0; n < order - k; n++) {
            const i32 Ctmp1 = C0[n + k + 1], Ctmp2 = C1[n];
            C0[n + k + 1] = s_smlawb(Ctmp1, shl32(Ctmp2, 1), rc_tmp_Q15);
            C1[n] = s_smlawb(Ctmp2, shl32(Ctmp1, 1), rc_tmp_Q15);
        }
    }
    for (; k < order; k++) rc_Q15[k] = 0;
    return imax(1, C1[0]);
}

CA_DEV void silk_k2a_dev(i32 *A_Q24, const i16 *rc_Q15, int order)                          // k2a_FIX.c:35-53
{
    i32 Atmp[SILK_MAX_LPC];
    for (int k = 0; k < order; k++) {
        for (int n = 0; n < k; n++) Atmp[n] = A_Q24[n];
        for (int n = 0; n < k; n++) A_Q24[n] = s_smlawb(A_Q24[n], shl32(Atmp[k - n - 1], 1), rc_Q15[k]);
        A_Q24[k] = (i32)(0u - (u32)shl32((i32)rc_Q15[k], 9));
    }
}

CA_DEV void silk_bwexpander_dev(i16 *ar, int d, i32 chirp_Q16)                              // bwexpander.c:35-51
{
    const i32 chirp_minus_one_Q16 = chirp_Q16 - 65536;
    for (int i = 0; i < d - 1; i++) {
        ar[i] = (i16)s_rshift_round(s_mulw(chirp_Q16, ar[i]), 16);
        chirp_Q16 += s_rshift_round(s_mulw(chirp_Q16, chirp_minus_one_Q16), 16);
    }
    ar[d - 1] = (i16)s_rshift_round(s_mulw(chirp_Q16, ar[d - 1]), 16);
}

// The usual geometries (16 kHz and 8 kHz input, 20 ms frames) as instances of the inlined body with the sampling rate and the subframe
// count constant (lengths, lag ranges and the subframe loops become constants); anything else through the generic instance.
template <class XA, class SCR>
CA_DEV int silk_pitch_analysis_core_dev(XA frame, SCR scr, int *pitch_out, int *lagIndex, int *contourIndex, int *LTPCorr_Q15, int prevLag,
                                        i32 search_thres1_Q16, int search_thres2_Q13, int Fs_kHz, int complexity, int nb_subfr)
{
    if (Fs_kHz == 16 && nb_subfr == 4)
        return silk_pitch_analysis_core_geom_dev(frame, scr, pitch_out, lagIndex, contourIndex, LTPCorr_Q15, prevLag, search_thres1_Q16, search_thres2_Q13, 16, complexity, 4);
    if (Fs_kHz == 8 && nb_subfr == 4)
        return silk_pitch_analysis_core_geom_dev(frame, scr, pitch_out, lagIndex, contourIndex, LTPCorr_Q15, prevLag, search_thres1_Q16, search_thres2_Q13, 8, complexity, 4);
    return silk_pitch_analysis_core_geom_dev(frame, scr, pitch_out, lagIndex, contourIndex, LTPCorr_Q15, prevLag, search_thres1_Q16, search_thres2_Q13, Fs_kHz, complexity, nb_subfr);
}

// LPC_analysis_filter.c:41-108, FIXED_POINT branch: out[0 .. d) = 0, out[ix] = SAT16(in[ix] + PSHR32(sum_m (-B[m]) * in[ix - 1 - m], 12))
template <class OUT, class XA>
CA_DEV void silk_LPC_analysis_filter_dev(OUT out, XA in, const i16 *B, int len, int d)
{
    // the d previous input samples travel in a register window: every sample is read once
    i32 nB[SILK_MAX_LPC], w[SILK_MAX_LPC];
    for (int j = 0; j < SILK_MAX_LPC; j++) { nB[j] = j < d ? (i32)(i16)(-(i32)B[j]) : 0; w[j] = j < d ? (i32)in[d - 1 - j] : 0; }
    for (int ix = d; ix < len; ix++) {
        i32 sum = 0;
#pragma unroll
        for (int m = 0; m < SILK_MAX_LPC; m++) sum = s_addw(sum, __mul24(nB[m], w[m]));
        const i32 xi = (i32)in[ix];
        const i32 v = xi + pshr32(sum, 12);
        out[ix] = (i16)(v > 32767 ? 32767 : (v < -32768 ? -32768 : v));
#pragma unroll
        for (int m = SILK_MAX_LPC - 1; m > 0; m--) w[m] = w[m - 1];
        w[0] = xi;
    }
    for (int j = 0; j < d; j++) out[j] = 0;
}

// silk_LPC_analysis_filter_dev with eight samples per memory access and a second copy of the first len2 output samples in out2 (the
// pitch estimator reads the whitened frame many times: from the caller's fast storage instead of from the output record)
template <class OUT, class OUT2, class XA>
CA_DEV void silk_LPC_analysis_filter_dup_dev(OUT out, OUT2 out2, int len2, XA in, const i16 *B, int len, int d)
{
    i32 nB[SILK_MAX_LPC], w[SILK_MAX_LPC];
    for (int j = 0; j < SILK_MAX_LPC; j++) { nB[j] = j < d ? (i32)(i16)(-(i32)B[j]) : 0; w[j] = 0; }
    int ix = 0;
    for (; ix + 8 <= len; ix += 8) {
        i32 xi[8], y[8];
        pe_load8(xi, in + ix);
#pragma unroll
        for (int u = 0; u < 8; u++) {
            i32 sum = 0;
#pragma unroll
            for (int m = 0; m < SILK_MAX_LPC; m++) sum = s_addw(sum, __mul24(nB[m], w[m]));
            const i32 v = xi[u] + pshr32(sum, 12);
            y[u] = ix + u < d ? 0 : (v > 32767 ? 32767 : (v < -32768 ? -32768 : v));        // out[0 .. d) = 0; the window fills meanwhile
#pragma unroll
            for (int m = SILK_MAX_LPC - 1; m > 0; m--) w[m] = w[m - 1];
            w[0] = xi[u];
        }
        pe_store8(out + ix, y);
        if (ix + 8 <= len2) pe_store8(out2 + ix, y);
        else for (int u = 0; u < 8; u++) if (ix + u < len2) out2[ix + u] = (i16)y[u];
    }
    for (; ix < len; ix++) {
        i32 sum = 0;
#pragma unroll
        for (int m = 0; m < SILK_MAX_LPC; m++) sum = s_addw(sum, __mul24(nB[m], w[m]));
        const i32 xi = (i32)in[ix];
        const i32 v = xi + pshr32(sum, 12);
        const i16 y = (i16)(ix < d ? 0 : (v > 32767 ? 32767 : (v < -32768 ? -32768 : v)));
        out[ix] = y;
        if (ix < len2) out2[ix] = y;
#pragma unroll
        for (int m = SILK_MAX_LPC - 1; m > 0; m--) w[m] = w[m - 1];
        w[0] = xi;
    }
}

struct PitchCfg {                                       // the psEnc fields the call reads
    int fs_kHz, nb_subfr, frame_length, ltp_mem_length, la_pitch, pitch_LPC_win_length, pitchEstimationLPCOrder, pitchEstimationComplexity,
        pitchEstimationThreshold_Q16, signalType, first_frame_after_reset, speech_activity_Q8, prevSignalType, input_tilt_Q15, prevLag,
        LTPCorr_Q15;
};

struct PitchOut {                                       // what it writes besides res[]
    int pitchL[4], lagIndex, contourIndex, LTPCorr_Q15, signalType, predGain_Q16;
};

// find_pitch_lags_FIX.c:83-103: reflection coefficients, prediction gain, bandwidth-expanded whitening filter from the autocorrelation
CA_DEV i32 pitch_whitening_filter_dev(i16 *A_Q12, const i32 *auto_corr, const int order)
{
    i32 A_Q24[SILK_MAX_LPC];
    i16 rc_Q15[SILK_MAX_LPC];
    const i32 res_nrg = silk_schur_dev(rc_Q15, auto_corr, order);
    const i32 predGain_Q16 = s_div32_varq(auto_corr[0], imax(res_nrg, 1), 16);
    silk_k2a_dev(A_Q24, rc_Q15, order);
    for (int i = 0; i < order; i++) {
        const i32 v = A_Q24[i] >> 12;
        A_Q12[i] = (i16)(v > 32767 ? 32767 : (v < -32768 ? -32768 : v));
    }
    silk_bwexpander_dev(A_Q12, order, 64881);                                               // FIND_PITCH_BANDWIDTH_EXPANSION Q16
    return predGain_Q16;
}

// x_buf: index 0 = x - ltp_mem_length (buf_len samples); res: buf_len samples out; ws / xs / scr (may be one array):
// scratch of pitch_LPC_win_length, pitch_LPC_win_length and (20 + 5 nb_subfr) fs_kHz samples in the caller's storage.
template <class XG, class RES, class SCR>
CA_DEV void silk_find_pitch_lags_dev(const PitchCfg &c, XG x_buf, RES res, SCR ws, SCR xs, SCR scr, PitchOut &o)
{
    const int buf_len = c.la_pitch + c.frame_length + c.ltp_mem_length, W = c.pitch_LPC_win_length, order = c.pitchEstimationLPCOrder;
    const XG xw = x_buf + (buf_len - W);
    silk_apply_sine_window_dev(ws, xw, 1, c.la_pitch);
    const int mid = W - (c.la_pitch << 1);
    {
        int i = 0;
        for (; i + 8 <= mid; i += 8) { i32 v[8]; pe_load8(v, xw + (c.la_pitch + i)); pe_store8(ws + (c.la_pitch + i), v); }
        for (; i < mid; i++) ws[c.la_pitch + i] = (i16)(i32)xw[c.la_pitch + i];
    }
    silk_apply_sine_window_dev(ws + (c.la_pitch + mid), xw + (c.la_pitch + mid), 2, c.la_pitch);
    i32 auto_corr[SILK_MAX_LPC + 1];
    i16 A_Q12[SILK_MAX_LPC];
    (void)silk_autocorr_dev(auto_corr, ws, xs, W, order + 1);
    auto_corr[0] = s_addw(s_smlawb(auto_corr[0], auto_corr[0], 66), 1);
    // the orders the complexity settings use (control_codec.c:333-404) as constants: Schur / k2a / bandwidth expansion unrolled, in registers
    if (order == 16) o.predGain_Q16 = pitch_whitening_filter_dev(A_Q12, auto_corr, 16);
    else if (order == 12) o.predGain_Q16 = pitch_whitening_filter_dev(A_Q12, auto_corr, 12);
    else if (order == 10) o.predGain_Q16 = pitch_whitening_filter_dev(A_Q12, auto_corr, 10);
    else if (order == 8) o.predGain_Q16 = pitch_whitening_filter_dev(A_Q12, auto_corr, 8);
    else o.predGain_Q16 = pitch_whitening_filter_dev(A_Q12, auto_corr, order);
    // the estimator's frame ((20 + 5 nb_subfr) ms of the whitened buffer) also goes to scr, where the estimator reads it (ws / xs are dead)
    const int core_len = (PE_LTP_MEM_LENGTH_MS + c.nb_subfr * PE_SUBFR_LENGTH_MS) * c.fs_kHz;
    silk_LPC_analysis_filter_dup_dev(res, scr, core_len, x_buf, A_Q12, buf_len, order);
    o.signalType = c.signalType;
    o.LTPCorr_Q15 = c.LTPCorr_Q15;
    if (c.signalType != 0 && c.first_frame_after_reset == 0) {
        i32 thrhld_Q13 = 4915;
        thrhld_Q13 = thrhld_Q13 + s_smulbb(-32, order);
        thrhld_Q13 = s_smlawb(thrhld_Q13, -209714, c.speech_activity_Q8);
        thrhld_Q13 = thrhld_Q13 + s_smulbb(-1228, c.prevSignalType >> 1);
        thrhld_Q13 = s_smlawb(thrhld_Q13, -1637, c.input_tilt_Q15);
        thrhld_Q13 = thrhld_Q13 > 32767 ? 32767 : (thrhld_Q13 < -32768 ? -32768 : thrhld_Q13);
        const int unvoiced = silk_pitch_analysis_core_dev(scr, scr, o.pitchL, &o.lagIndex, &o.contourIndex, &o.LTPCorr_Q15, c.prevLag,
                                                          c.pitchEstimationThreshold_Q16, (int)thrhld_Q13, c.fs_kHz, c.pitchEstimationComplexity,
                                                          c.nb_subfr);
        o.signalType = unvoiced ? 1 : 2;
    } else {
        for (int k = 0; k < 4; k++) o.pitchL[k] = 0;
        o.lagIndex = 0;
        o.contourIndex = 0;
        o.LTPCorr_Q15 = 0;
    }
}

}  // namespace ca

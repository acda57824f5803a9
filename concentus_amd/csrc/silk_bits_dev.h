// silk_bits_dev.h -- the entropy-coding side of a SILK frame (eighth slice): side information and excitation
// onto the Opus range coder, i.e. what silk_encode_frame_FIX (opus-fix/silk/fixed/encode_frame_FIX.c) runs after the quantiser.
//
//   silk_encode_indices      opus-fix/silk/encode_indices.c:36-193
//   silk_encode_pulses       opus-fix/silk/encode_pulses.c:64-206 (+ combine_and_check :38-61)
//   silk_shell_encoder       opus-fix/silk/shell_coder.c:78-112
//   silk_encode_signs        opus-fix/silk/code_signs.c:44-79
//   silk_NLSF_unpack         (silk_nlsf_dev.h)
//   ec_enc_icdf              (rangecoder.h; opus-fix/celt/entenc.c:279-292)
//
// One lane owns one frame and its range coder (the build defines CA_LANE_FRAME: the coder's byte writes are per lane).
#pragma once
#include "rangecoder.h"
#include "silk_nlsf_dev.h"
#include "silk_entropy_tables.h"

namespace ca {

enum { SHELL_FRAME = 16, SILK_MAX_PULSES = 16, N_RATE_LEVELS = 10, MAX_SHELL_BLOCKS = 20 };

struct SilkIndices {            // SideInfoIndices (opus-fix/silk/structs.h:113-127)
    i8 GainsIndices[4], LTPIndex[4], NLSFIndices[SILK_MAX_LPC + 1];
    int lagIndex, contourIndex, signalType, quantOffsetType, NLSFInterpCoef_Q2, PERIndex, LTP_scaleIndex, Seed;
};

// encode_indices.c:36-193 for encode_LBRR == 0; ec_prev* are psEncC->ec_prevSignalType / ec_prevLagIndex (I/O)
CA_DEV void silk_encode_indices_dev(RangeEnc &ec, const SilkIndices &ix, int nb_subfr, int fs_kHz, int order, int condCoding,
                                    int &ec_prevSignalType, int &ec_prevLagIndex, const NlsfTablesLds *tables = nullptr)
{
    const int typeOffset = 2 * ix.signalType + ix.quantOffsetType;
    if (typeOffset >= 2) ec_enc_icdf(ec, typeOffset - 2, SILK_type_offset_VAD_iCDF, 8);
    else ec_enc_icdf(ec, typeOffset, SILK_type_offset_no_VAD_iCDF, 8);
    if (condCoding == 2) {                                                                  // CODE_CONDITIONALLY
        ec_enc_icdf(ec, ix.GainsIndices[0], SILK_delta_gain_iCDF, 8);
    } else {
        ec_enc_icdf(ec, ix.GainsIndices[0] >> 3, &SILK_gain_iCDF[ix.signalType * 8], 8);
        ec_enc_icdf(ec, ix.GainsIndices[0] & 7, SILK_uniform8_iCDF, 8);
    }
    for (int i = 1; i < nb_subfr; i++) ec_enc_icdf(ec, ix.GainsIndices[i], SILK_delta_gain_iCDF, 8);
    // NLSFs
    const NlsfCB cb = nlsf_codebook(order, tables);
    ec_enc_icdf(ec, ix.NLSFIndices[0], &cb.CB1_iCDF[(ix.signalType >> 1) * cb.nVectors], 8);
    i16 ec_ix[SILK_MAX_LPC];
    u8 pred_Q8[SILK_MAX_LPC];
    silk_NLSF_unpack_dev(ec_ix, pred_Q8, cb, ix.NLSFIndices[0]);
    for (int i = 0; i < cb.order; i++) {
        const int v = ix.NLSFIndices[i + 1];
        if (v >= NLSF_MAX_AMP) {
            ec_enc_icdf(ec, 2 * NLSF_MAX_AMP, &cb.ec_iCDF[ec_ix[i]], 8);
            ec_enc_icdf(ec, v - NLSF_MAX_AMP, SILK_NLSF_EXT_iCDF, 8);
        } else if (v <= -NLSF_MAX_AMP) {
            ec_enc_icdf(ec, 0, &cb.ec_iCDF[ec_ix[i]], 8);
            ec_enc_icdf(ec, -v - NLSF_MAX_AMP, SILK_NLSF_EXT_iCDF, 8);
        } else {
            ec_enc_icdf(ec, v + NLSF_MAX_AMP, &cb.ec_iCDF[ec_ix[i]], 8);
        }
    }
    if (nb_subfr == 4) ec_enc_icdf(ec, ix.NLSFInterpCoef_Q2, SILK_NLSF_interpolation_factor_iCDF, 8);
    if (ix.signalType == 2) {
        int encode_absolute_lagIndex = 1;
        if (condCoding == 2 && ec_prevSignalType == 2) {
            int delta_lagIndex = ix.lagIndex - ec_prevLagIndex;
            if (delta_lagIndex < -8 || delta_lagIndex > 11) {
                delta_lagIndex = 0;
            } else {
                delta_lagIndex += 9;
                encode_absolute_lagIndex = 0;
            }
            ec_enc_icdf(ec, delta_lagIndex, SILK_pitch_delta_iCDF, 8);
        }
        if (encode_absolute_lagIndex) {
            const int half = fs_kHz >> 1;
            const i32 pitch_high_bits = ix.lagIndex / half;
            const i32 pitch_low_bits = ix.lagIndex - s_smulbb(pitch_high_bits, half);
            ec_enc_icdf(ec, pitch_high_bits, SILK_pitch_lag_iCDF, 8);
            // psEncC->pitch_lag_low_bits_iCDF (control_codec.c:296-304): uniform 8 / 6 / 4 at 16 / 12 / 8 kHz
            ec_enc_icdf(ec, pitch_low_bits, fs_kHz == 16 ? SILK_uniform8_iCDF : fs_kHz == 12 ? SILK_uniform6_iCDF : SILK_uniform4_iCDF, 8);
        }
        ec_prevLagIndex = (i16)ix.lagIndex;
        // psEncC->pitch_contour_iCDF (control_codec.c:219-276)
        const u8 *contour = nb_subfr == 4 ? (fs_kHz == 8 ? SILK_pitch_contour_NB_iCDF : SILK_pitch_contour_iCDF)
                                          : (fs_kHz == 8 ? SILK_pitch_contour_10_ms_NB_iCDF : SILK_pitch_contour_10_ms_iCDF);
        ec_enc_icdf(ec, ix.contourIndex, contour, 8);
        ec_enc_icdf(ec, ix.PERIndex, SILK_LTP_per_index_iCDF, 8);
        const u8 *ltp_icdf = &SILK_LTP_gain_iCDF[ix.PERIndex == 0 ? 0 : ix.PERIndex == 1 ? 8 : 24];
        for (int k = 0; k < nb_subfr; k++) ec_enc_icdf(ec, ix.LTPIndex[k], ltp_icdf, 8);
        if (condCoding == 0) ec_enc_icdf(ec, ix.LTP_scaleIndex, SILK_LTPscale_iCDF, 8);      // CODE_INDEPENDENTLY
    }
    ec_prevSignalType = ix.signalType;
    ec_enc_icdf(ec, ix.Seed, SILK_uniform4_iCDF, 8);
}

CA_DEV void silk_encode_split(RangeEnc &ec, int p_child1, int p, const u8 *shell_table)     // shell_coder.c:46-55
{
    if (p > 0) ec_enc_icdf(ec, p_child1, &shell_table[SILK_shell_code_table_offsets[p]], 8);
}

CA_DEV void silk_shell_encoder_dev(RangeEnc &ec, const u8 *pulses0)                         // shell_coder.c:78-112
{
    int p1[8], p2[4], p3[2], p4;
    for (int k = 0; k < 8; k++) p1[k] = pulses0[2 * k] + pulses0[2 * k + 1];
    for (int k = 0; k < 4; k++) p2[k] = p1[2 * k] + p1[2 * k + 1];
    for (int k = 0; k < 2; k++) p3[k] = p2[2 * k] + p2[2 * k + 1];
    p4 = p3[0] + p3[1];
    silk_encode_split(ec, p3[0], p4, SILK_shell_code_table3);
    silk_encode_split(ec, p2[0], p3[0], SILK_shell_code_table2);
    silk_encode_split(ec, p1[0], p2[0], SILK_shell_code_table1);
    silk_encode_split(ec, pulses0[0], p1[0], SILK_shell_code_table0);
    silk_encode_split(ec, pulses0[2], p1[1], SILK_shell_code_table0);
    silk_encode_split(ec, p1[2], p2[1], SILK_shell_code_table1);
    silk_encode_split(ec, pulses0[4], p1[2], SILK_shell_code_table0);
    silk_encode_split(ec, pulses0[6], p1[3], SILK_shell_code_table0);
    silk_encode_split(ec, p2[2], p3[1], SILK_shell_code_table2);
    silk_encode_split(ec, p1[4], p2[2], SILK_shell_code_table1);
    silk_encode_split(ec, pulses0[8], p1[4], SILK_shell_code_table0);
    silk_encode_split(ec, pulses0[10], p1[5], SILK_shell_code_table0);
    silk_encode_split(ec, p1[6], p2[3], SILK_shell_code_table1);
    silk_encode_split(ec, pulses0[12], p1[6], SILK_shell_code_table0);
    silk_encode_split(ec, pulses0[14], p1[7], SILK_shell_code_table0);
}

// encode_pulses.c:38-61: pairwise sums of `len` outputs; returns 1 if a sum exceeds max_pulses (outputs up to there are written)
CA_DEV int silk_combine_and_check(int *out, const int *in, int max_pulses, int len)
{
    for (int k = 0; k < len; k++) {
        const int sum = in[2 * k] + in[2 * k + 1];
        if (sum > max_pulses) return 1;
        out[k] = sum;
    }
    return 0;
}

// encode_pulses.c:64-206; pulses: frame_length quantisation indices (frame_length a multiple of 16 here: 8 / 16 kHz);
// absq: scratch of frame_length bytes
template <class PA>
CA_DEV void silk_encode_pulses_dev(RangeEnc &ec, int signalType, int quantOffsetType, PA pulses, int frame_length, u8 *absq)
{
    const int iter = frame_length >> 4;
    int sum_pulses[MAX_SHELL_BLOCKS], nRshifts[MAX_SHELL_BLOCKS];
    (void)absq;
    // |q| of a shell block, down-shifted: one 16-byte access of the pulses per block and pass (the passes below re-derive what the
    // reference keeps in abs_pulses[]: a block's magnitudes are needed shifted by its final nRshifts only)
    struct B16 { i8 b[SHELL_FRAME]; };
    auto block_abs = [&](int i, int rshift, u8 *a16) {
        B16 blk;
        __builtin_memcpy(&blk, &pulses[i * SHELL_FRAME], sizeof(blk));
#pragma unroll
        for (int k = 0; k < SHELL_FRAME; k++) { const int q = blk.b[k]; a16[k] = (u8)((q < 0 ? -q : q) >> rshift); }
    };
    for (int i = 0; i < iter; i++) {
        u8 ap[SHELL_FRAME];
        block_abs(i, 0, ap);
        nRshifts[i] = 0;
        while (1) {
            int a16[16], comb[8];
            for (int k = 0; k < 8; k++) comb[k] = 0;
            for (int k = 0; k < 16; k++) a16[k] = ap[k];
            int scale_down = silk_combine_and_check(comb, a16, SILK_max_pulses_table[0], 8);
            scale_down += silk_combine_and_check(comb, comb, SILK_max_pulses_table[1], 4);
            scale_down += silk_combine_and_check(comb, comb, SILK_max_pulses_table[2], 2);
            scale_down += silk_combine_and_check(&sum_pulses[i], comb, SILK_max_pulses_table[3], 1);
            if (!scale_down) break;
            nRshifts[i]++;
            for (int k = 0; k < SHELL_FRAME; k++) ap[k] = (u8)(ap[k] >> 1);
        }
    }
    // rate level (:135-152)
    i32 minSumBits_Q5 = 0x7FFFFFFF;
    int RateLevelIndex = 0;
    for (int k = 0; k < N_RATE_LEVELS - 1; k++) {
        const u8 *nBits_ptr = &SILK_pulses_per_block_BITS_Q5[k * 18];
        i32 sumBits_Q5 = SILK_rate_levels_BITS_Q5[(signalType >> 1) * 9 + k];
        for (int i = 0; i < iter; i++) sumBits_Q5 += nRshifts[i] > 0 ? nBits_ptr[SILK_MAX_PULSES + 1] : nBits_ptr[sum_pulses[i]];
        if (sumBits_Q5 < minSumBits_Q5) { minSumBits_Q5 = sumBits_Q5; RateLevelIndex = k; }
    }
    ec_enc_icdf(ec, RateLevelIndex, &SILK_rate_levels_iCDF[(signalType >> 1) * 9], 8);
    // sum-weighted-pulses encoding (:154-170)
    const u8 *cdf_ptr = &SILK_pulses_per_block_iCDF[RateLevelIndex * 18], *cdf_last = &SILK_pulses_per_block_iCDF[(N_RATE_LEVELS - 1) * 18];
    for (int i = 0; i < iter; i++) {
        if (nRshifts[i] == 0) {
            ec_enc_icdf(ec, sum_pulses[i], cdf_ptr, 8);
        } else {
            ec_enc_icdf(ec, SILK_MAX_PULSES + 1, cdf_ptr, 8);
            for (int k = 0; k < nRshifts[i] - 1; k++) ec_enc_icdf(ec, SILK_MAX_PULSES + 1, cdf_last, 8);
            ec_enc_icdf(ec, sum_pulses[i], cdf_last, 8);
        }
    }
    // shell encoding (:172-179)
    for (int i = 0; i < iter; i++)
        if (sum_pulses[i] > 0) {
            u8 ap[SHELL_FRAME];
            block_abs(i, nRshifts[i], ap);
            silk_shell_encoder_dev(ec, ap);
        }
    // LSB encoding (:181-199)
    for (int i = 0; i < iter; i++) {
        if (nRshifts[i] > 0) {
            const int nLS = nRshifts[i] - 1;
            B16 blk;
            __builtin_memcpy(&blk, &pulses[i * SHELL_FRAME], sizeof(blk));
#pragma unroll
            for (int k = 0; k < SHELL_FRAME; k++) {
                const int q = blk.b[k];
                const i32 abs_q = (i8)(q < 0 ? -q : q);
                for (int j = nLS; j > 0; j--) ec_enc_icdf(ec, (abs_q >> j) & 1, SILK_lsb_iCDF, 8);
                ec_enc_icdf(ec, abs_q & 1, SILK_lsb_iCDF, 8);
            }
        }
    }
    // signs (code_signs.c:44-79)
    {
        u8 icdf[2];
        icdf[1] = 0;
        const u8 *icdf_ptr = &SILK_sign_iCDF[s_smulbb(7, quantOffsetType + (signalType << 1))];
        const int length = (frame_length + SHELL_FRAME / 2) >> 4;
        for (int i = 0; i < length; i++) {
            const int p = sum_pulses[i];
            if (p > 0) {
                icdf[0] = icdf_ptr[imin(p & 0x1F, 6)];
                B16 blk;
                __builtin_memcpy(&blk, &pulses[i * SHELL_FRAME], sizeof(blk));
#pragma unroll
                for (int j = 0; j < SHELL_FRAME; j++) {
                    const int q = blk.b[j];
                    if (q != 0) ec_enc_icdf(ec, (q >> 15) + 1, icdf, 8);                    // silk_enc_map
                }
            }
        }
    }
}

}  // namespace ca

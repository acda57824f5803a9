// silk_nlsf_dev.h -- NLSF quantisation and the quantised-LPC residual energies of the SILK encoder (second
// slice): what silk_find_pred_coefs_FIX (opus-fix/silk/fixed/find_pred_coefs_FIX.c:139-143) runs after silk_find_LPC_FIX.
//
//   silk_process_NLSFs                 opus-fix/silk/process_NLSFs.c:35-106
//   silk_NLSF_encode                   opus-fix/silk/NLSF_encode.c:38-157
//   silk_NLSF_VQ                       opus-fix/silk/NLSF_VQ.c:35-68
//   silk_NLSF_del_dec_quant            opus-fix/silk/NLSF_del_dec_quant.c:35-217
//   silk_NLSF_unpack                   opus-fix/silk/NLSF_unpack.c:35-55
//   silk_NLSF_decode                   opus-fix/silk/NLSF_decode.c:35-101
//   silk_NLSF_VQ_weights_laroia        opus-fix/silk/NLSF_VQ_weights_laroia.c:41-80
//   silk_NLSF_stabilize                opus-fix/silk/NLSF_stabilize.c:46-142
//   silk_interpolate                   opus-fix/silk/interpolate.c:35-51
//   silk_insertion_sort_increasing     opus-fix/silk/sort.c:38-84
//   silk_lin2log                       opus-fix/silk/lin2log.c:35-45
//   silk_residual_energy_FIX           opus-fix/silk/fixed/residual_energy_FIX.c:37-98
//
// One lane owns one record; everything here is a short serial recurrence (a 16-step trellis, a 32-entry insertion sort, a
// root-free polynomial expansion). Integer arithmetic throughout, results bit-exact.
#pragma once
#include "silk_lpc_dev.h"
#include "silk_nlsf_tables.h"

namespace ca {

enum { NLSF_W_Q = 2, NLSF_MAX_SURVIVORS = 32, NLSF_MAX_AMP = 4, NLSF_MAX_AMP_EXT = 10, NLSF_DD_STATES = 4, NLSF_DD_STATES_LOG2 = 2,
       NLSF_LEVEL_ADJ_Q10 = 102 /* SILK_FIX_CONST(0.1, 10) */ };

struct NlsfCB {                 // cf. silk_NLSF_CB_struct (structs.h:83-95)
    int nVectors, order, quantStepSize_Q16, invQuantStepSize_Q6;
    const u8 *CB1_Q8, *CB1_iCDF, *pred_Q8, *ec_sel, *ec_iCDF, *ec_Rates_Q5;
    const i16 *deltaMin_Q15;
};

// the codebook the encoder selects (control_codec.c: 16 kHz -> order 16, silk_NLSF_CB_WB; 8 / 12 kHz -> order 10, NB_MB)
CA_DEV NlsfCB nlsf_codebook(int order)
{
    NlsfCB cb;
    if (order == 16) {
        cb.nVectors = SILK_NLSF_WB_NVECTORS; cb.order = SILK_NLSF_WB_ORDER;
        cb.quantStepSize_Q16 = SILK_NLSF_WB_QUANT_STEP_Q16; cb.invQuantStepSize_Q6 = SILK_NLSF_WB_INV_QUANT_STEP_Q6;
        cb.CB1_Q8 = SILK_NLSF_WB_CB1_Q8; cb.CB1_iCDF = SILK_NLSF_WB_CB1_iCDF; cb.pred_Q8 = SILK_NLSF_WB_pred_Q8;
        cb.ec_sel = SILK_NLSF_WB_ec_sel; cb.ec_iCDF = SILK_NLSF_WB_ec_iCDF; cb.ec_Rates_Q5 = SILK_NLSF_WB_ec_Rates_Q5;
        cb.deltaMin_Q15 = SILK_NLSF_WB_deltaMin_Q15;
    } else {
        cb.nVectors = SILK_NLSF_NB_MB_NVECTORS; cb.order = SILK_NLSF_NB_MB_ORDER;
        cb.quantStepSize_Q16 = SILK_NLSF_NB_MB_QUANT_STEP_Q16; cb.invQuantStepSize_Q6 = SILK_NLSF_NB_MB_INV_QUANT_STEP_Q6;
        cb.CB1_Q8 = SILK_NLSF_NB_MB_CB1_Q8; cb.CB1_iCDF = SILK_NLSF_NB_MB_CB1_iCDF; cb.pred_Q8 = SILK_NLSF_NB_MB_pred_Q8;
        cb.ec_sel = SILK_NLSF_NB_MB_ec_sel; cb.ec_iCDF = SILK_NLSF_NB_MB_ec_iCDF; cb.ec_Rates_Q5 = SILK_NLSF_NB_MB_ec_Rates_Q5;
        cb.deltaMin_Q15 = SILK_NLSF_NB_MB_deltaMin_Q15;
    }
    return cb;
}

// A workgroup's LDS copy of both codebooks (2.1 KB): the VQ, the entropy-table look-ups of the trellis and the decode read them
// per lane at data-dependent addresses, a global / L2 round trip each otherwise.
struct NlsfTablesLds {
    u8 wb_cb1[512], wb_icdf[64], wb_pred[32], wb_ec_sel[256], wb_ec_icdf[72], wb_rates[72];
    u8 nb_cb1[320], nb_icdf[64], nb_pred[20], nb_ec_sel[160], nb_ec_icdf[72], nb_rates[72];
    i16 wb_delta[18], nb_delta[12];
};

CA_DEV void nlsf_stage_tables(NlsfTablesLds &T, int tid, int nthreads)
{
#define CA_NLSF_COPY(dst, src, n) for (int k = tid; k < (n); k += nthreads) (dst)[k] = (src)[k]
    CA_NLSF_COPY(T.wb_cb1, SILK_NLSF_WB_CB1_Q8, 512); CA_NLSF_COPY(T.wb_icdf, SILK_NLSF_WB_CB1_iCDF, 64); CA_NLSF_COPY(T.wb_pred, SILK_NLSF_WB_pred_Q8, 30);
    CA_NLSF_COPY(T.wb_ec_sel, SILK_NLSF_WB_ec_sel, 256); CA_NLSF_COPY(T.wb_ec_icdf, SILK_NLSF_WB_ec_iCDF, 72); CA_NLSF_COPY(T.wb_rates, SILK_NLSF_WB_ec_Rates_Q5, 72);
    CA_NLSF_COPY(T.wb_delta, SILK_NLSF_WB_deltaMin_Q15, 17);
    CA_NLSF_COPY(T.nb_cb1, SILK_NLSF_NB_MB_CB1_Q8, 320); CA_NLSF_COPY(T.nb_icdf, SILK_NLSF_NB_MB_CB1_iCDF, 64); CA_NLSF_COPY(T.nb_pred, SILK_NLSF_NB_MB_pred_Q8, 18);
    CA_NLSF_COPY(T.nb_ec_sel, SILK_NLSF_NB_MB_ec_sel, 160); CA_NLSF_COPY(T.nb_ec_icdf, SILK_NLSF_NB_MB_ec_iCDF, 72); CA_NLSF_COPY(T.nb_rates, SILK_NLSF_NB_MB_ec_Rates_Q5, 72);
    CA_NLSF_COPY(T.nb_delta, SILK_NLSF_NB_MB_deltaMin_Q15, 11);
#undef CA_NLSF_COPY
}

CA_DEV NlsfCB nlsf_codebook(int order, const NlsfTablesLds *T)
{
    NlsfCB cb = nlsf_codebook(order);
    if (!T) return cb;
    if (order == 16) {
        cb.CB1_Q8 = T->wb_cb1; cb.CB1_iCDF = T->wb_icdf; cb.pred_Q8 = T->wb_pred; cb.ec_sel = T->wb_ec_sel; cb.ec_iCDF = T->wb_ec_icdf;
        cb.ec_Rates_Q5 = T->wb_rates; cb.deltaMin_Q15 = T->wb_delta;
    } else {
        cb.CB1_Q8 = T->nb_cb1; cb.CB1_iCDF = T->nb_icdf; cb.pred_Q8 = T->nb_pred; cb.ec_sel = T->nb_ec_sel; cb.ec_iCDF = T->nb_ec_icdf;
        cb.ec_Rates_Q5 = T->nb_rates; cb.deltaMin_Q15 = T->nb_delta;
    }
    return cb;
}

CA_DEV i32 s_lin2log(i32 inLin)                                                            // lin2log.c:35-45, Inlines.h:56-66
{
    const int lz = s_clz32(inLin);
    const int rot = 24 - lz;                                                                // silk_ROR32, either direction
    const u32 x = (u32)inLin;
    const u32 r = rot == 0 ? x : rot > 0 ? ((x >> rot) | (x << (32 - rot))) : ((x << -rot) | (x >> (32 + rot)));
    const i32 frac_Q7 = (i32)(r & 0x7f);
    return shl32(31 - lz, 7) + s_smlawb(frac_Q7, frac_Q7 * (128 - frac_Q7), 179);
}

CA_DEV void silk_NLSF_stabilize_dev(i16 *NLSF_Q15, const i16 *NDeltaMin_Q15, int L)         // NLSF_stabilize.c:46-142
{
    int loops;
    for (loops = 0; loops < 20; loops++) {
        i32 min_diff = NLSF_Q15[0] - NDeltaMin_Q15[0];
        int I = 0;
        for (int i = 1; i <= L - 1; i++) {
            const i32 diff = NLSF_Q15[i] - (NLSF_Q15[i - 1] + NDeltaMin_Q15[i]);
            if (diff < min_diff) { min_diff = diff; I = i; }
        }
        const i32 diff = (1 << 15) - (NLSF_Q15[L - 1] + NDeltaMin_Q15[L]);
        if (diff < min_diff) { min_diff = diff; I = L; }
        if (min_diff >= 0) return;
        if (I == 0) {
            NLSF_Q15[0] = NDeltaMin_Q15[0];
        } else if (I == L) {
            NLSF_Q15[L - 1] = (i16)((1 << 15) - NDeltaMin_Q15[L]);
        } else {
            i32 min_center = 0, max_center = 1 << 15;
            for (int k = 0; k < I; k++) min_center += NDeltaMin_Q15[k];
            min_center += NDeltaMin_Q15[I] >> 1;
            for (int k = L; k > I; k--) max_center -= NDeltaMin_Q15[k];
            max_center -= NDeltaMin_Q15[I] >> 1;
            const i16 center = (i16)s_limit(s_rshift_round((i32)NLSF_Q15[I - 1] + (i32)NLSF_Q15[I], 1), min_center, max_center);
            NLSF_Q15[I - 1] = (i16)(center - (NDeltaMin_Q15[I] >> 1));
            NLSF_Q15[I] = (i16)(NLSF_Q15[I - 1] + NDeltaMin_Q15[I]);
        }
    }
    // fall back (NLSF_stabilize.c:120-141): sort, then enforce the minimum distances from both ends
    for (int i = 1; i < L; i++) {                                                           // sort.c:134-154
        const i16 value = NLSF_Q15[i];
        int j;
        for (j = i - 1; j >= 0 && value < NLSF_Q15[j]; j--) NLSF_Q15[j + 1] = NLSF_Q15[j];
        NLSF_Q15[j + 1] = value;
    }
    NLSF_Q15[0] = (i16)imax(NLSF_Q15[0], NDeltaMin_Q15[0]);
    for (int i = 1; i < L; i++) NLSF_Q15[i] = (i16)imax(NLSF_Q15[i], NLSF_Q15[i - 1] + NDeltaMin_Q15[i]);
    NLSF_Q15[L - 1] = (i16)imin(NLSF_Q15[L - 1], (1 << 15) - NDeltaMin_Q15[L]);
    for (int i = L - 2; i >= 0; i--) NLSF_Q15[i] = (i16)imin(NLSF_Q15[i], NLSF_Q15[i + 1] - NDeltaMin_Q15[i + 1]);
}

CA_DEV void silk_NLSF_VQ_weights_laroia_dev(i16 *W, const i16 *NLSF_Q15, int D)             // NLSF_VQ_weights_laroia.c:41-80
{
    const i32 one = (i32)1 << (15 + NLSF_W_Q);
    i32 tmp1 = one / imax(NLSF_Q15[0], 1);
    i32 tmp2 = one / imax(NLSF_Q15[1] - NLSF_Q15[0], 1);
    W[0] = (i16)imin(tmp1 + tmp2, 32767);
#pragma unroll
    for (int k = 1; k < D - 1; k += 2) {
        tmp1 = one / imax(NLSF_Q15[k + 1] - NLSF_Q15[k], 1);
        W[k] = (i16)imin(tmp1 + tmp2, 32767);
        tmp2 = one / imax(NLSF_Q15[k + 2] - NLSF_Q15[k + 1], 1);
        W[k + 1] = (i16)imin(tmp1 + tmp2, 32767);
    }
    tmp1 = one / imax((1 << 15) - NLSF_Q15[D - 1], 1);
    W[D - 1] = (i16)imin(tmp1 + tmp2, 32767);
}

CA_DEV void silk_interpolate_dev(i16 *xi, const i16 *x0, const i16 *x1, int ifact_Q2, int d)  // interpolate.c:35-51
{
    for (int i = 0; i < d; i++) xi[i] = (i16)(x0[i] + (s_smulbb(x1[i] - x0[i], ifact_Q2) >> 2));
}

// K smallest of a[0..L) in increasing order, ties in index order (sort.c:38-84)
CA_DEV void silk_insertion_sort_increasing_dev(i32 *a, int *idx, int L, int K)
{
    for (int i = 0; i < K; i++) idx[i] = i;
    for (int i = 1; i < K; i++) {
        const i32 value = a[i];
        int j;
        for (j = i - 1; j >= 0 && value < a[j]; j--) { a[j + 1] = a[j]; idx[j + 1] = idx[j]; }
        a[j + 1] = value;
        idx[j + 1] = i;
    }
    for (int i = K; i < L; i++) {
        const i32 value = a[i];
        if (value < a[K - 1]) {
            int j;
            for (j = K - 2; j >= 0 && value < a[j]; j--) { a[j + 1] = a[j]; idx[j + 1] = idx[j]; }
            a[j + 1] = value;
            idx[j + 1] = i;
        }
    }
}

CA_DEV void silk_NLSF_unpack_dev(i16 *ec_ix, u8 *pred_Q8, const NlsfCB &cb, int CB1_index)   // NLSF_unpack.c:35-55
{
    const u8 *ec_sel_ptr = &cb.ec_sel[CB1_index * cb.order / 2];
#pragma unroll
    for (int i = 0; i < cb.order; i += 2) {
        const int entry = *ec_sel_ptr++;
        ec_ix[i] = (i16)(((entry >> 1) & 7) * (2 * NLSF_MAX_AMP + 1));
        pred_Q8[i] = cb.pred_Q8[i + (entry & 1) * (cb.order - 1)];
        ec_ix[i + 1] = (i16)(((entry >> 5) & 7) * (2 * NLSF_MAX_AMP + 1));
        pred_Q8[i + 1] = cb.pred_Q8[i + ((entry >> 4) & 1) * (cb.order - 1) + 1];
    }
}

// NLSF_del_dec_quant.c:35-217. ind[][] rows are written for every coefficient the moment it is visited, so the
// whole-row copy of the reference (:183, which also moves not-yet-written bytes) moves the same live bytes as the
// copy of the visited suffix done here.
#if defined(CA_HOST_EMU)
#define CA_NLSF_MEMBER inline
#else
#define CA_NLSF_MEMBER __device__ __forceinline__
#endif
// The trellis' survivor state: index paths, previous outputs, rate-distortion values. Every access is a dependent step of the search
// (indexed by state and coefficient at run time), so where it lives sets the pace: local arrays (private memory on the device) by
// default, a lane's column of an LDS block [slot][64 lanes] in the kernels (NlsfTrellisCol).
struct NlsfTrellisLocal {
    i32 v[28];
    i16 path[NLSF_DD_STATES * SILK_MAX_LPC];
    CA_NLSF_MEMBER i16 &ind(int j, int i) { return path[j * SILK_MAX_LPC + i]; }
    CA_NLSF_MEMBER i32 &prev_out(int j) { return v[j]; }
    CA_NLSF_MEMBER i32 &RD(int j) { return v[8 + j]; }
    CA_NLSF_MEMBER i32 &RD_min(int j) { return v[16 + j]; }
    CA_NLSF_MEMBER i32 &RD_max(int j) { return v[20 + j]; }
    CA_NLSF_MEMBER i32 &sort(int j) { return v[24 + j]; }
};
enum { NLSF_TRELLIS_SLOTS16 = NLSF_DD_STATES * SILK_MAX_LPC + 2 * 28 };     // 64 16-bit slots + 28 32-bit slots per lane
struct NlsfTrellisCol {
    i16 *p16;                                            // -> this lane's 16-bit column (slot stride 64)
    i32 *p32;                                            // -> this lane's 32-bit column behind it
    CA_NLSF_MEMBER i16 &ind(int j, int i) { return p16[(j * SILK_MAX_LPC + i) * 64]; }
    CA_NLSF_MEMBER i32 &prev_out(int j) { return p32[j * 64]; }
    CA_NLSF_MEMBER i32 &RD(int j) { return p32[(8 + j) * 64]; }
    CA_NLSF_MEMBER i32 &RD_min(int j) { return p32[(16 + j) * 64]; }
    CA_NLSF_MEMBER i32 &RD_max(int j) { return p32[(20 + j) * 64]; }
    CA_NLSF_MEMBER i32 &sort(int j) { return p32[(24 + j) * 64]; }
    // block: NLSF_TRELLIS_SLOTS16 * 64 16-bit words of LDS, 4-byte aligned; lane: 0 .. 63
    static CA_NLSF_MEMBER NlsfTrellisCol at(i16 *block, int lane)
    {
        NlsfTrellisCol t;
        t.p16 = block + lane;
        t.p32 = reinterpret_cast<i32 *>(block + NLSF_DD_STATES * SILK_MAX_LPC * 64) + lane;
        return t;
    }
};

// element idx (run-time, 0 .. 3) of a four-entry register array: compare chains instead of memory
CA_DEV i32 nlsf_get4(const i32 *a, int idx) { return idx == 0 ? a[0] : idx == 1 ? a[1] : idx == 2 ? a[2] : a[3]; }
CA_DEV void nlsf_set4(i32 *a, int idx, i32 v) { a[0] = idx == 0 ? v : a[0]; a[1] = idx == 1 ? v : a[1]; a[2] = idx == 2 ? v : a[2]; a[3] = idx == 3 ? v : a[3]; }

// One coefficient of the trellis with NS (1, 2 or 4: a compile-time constant) live states: every index into the rate-distortion
// values / previous outputs is then a constant and the two arrays are registers; only the index paths (T.ind: addressed by
// coefficient and, in the survivor exchange, by run-time state) stay in the trellis memory.
template <int NS, class TM>
CA_DEV void nlsf_trellis_step(TM &T, i32 *RD /*[8]*/, i32 *po /*[8]*/, const int i, const int order, const bool last, const int in_Q10, const i32 w_Q5i,
                              const i32 pred_coef_Q16, const u8 *rates_Q5, const int quant_step_size_Q16, const i32 inv_quant_step_size_Q6,
                              const i32 mu_Q20)
{
#pragma unroll
    for (int j = 0; j < NS; j++) {
        const int pred_Q10 = s_smulwb(pred_coef_Q16, po[j]);
        const int res_Q10 = in_Q10 - pred_Q10;
        int ind_tmp = s_smulwb(inv_quant_step_size_Q6, res_Q10);
        ind_tmp = s_limit(ind_tmp, -NLSF_MAX_AMP_EXT, NLSF_MAX_AMP_EXT - 1);
        T.ind(j, i) = (i8)ind_tmp;
        // out0 / out1: the two reconstruction levels around the residual (the reference tabulates them per call, :61-79)
        int out0_Q10 = shl32(ind_tmp, 10), out1_Q10 = out0_Q10 + 1024;
        if (ind_tmp > 0) { out0_Q10 -= NLSF_LEVEL_ADJ_Q10; out1_Q10 -= NLSF_LEVEL_ADJ_Q10; }
        else if (ind_tmp == 0) { out1_Q10 -= NLSF_LEVEL_ADJ_Q10; }
        else if (ind_tmp == -1) { out0_Q10 += NLSF_LEVEL_ADJ_Q10; }
        else { out0_Q10 += NLSF_LEVEL_ADJ_Q10; out1_Q10 += NLSF_LEVEL_ADJ_Q10; }
        out0_Q10 = s_smulwb(out0_Q10, quant_step_size_Q16) + pred_Q10;
        out1_Q10 = s_smulwb(out1_Q10, quant_step_size_Q16) + pred_Q10;
        po[j] = (i16)out0_Q10;
        po[j + NS] = (i16)out1_Q10;
        int rate0_Q5, rate1_Q5;
        if (ind_tmp + 1 >= NLSF_MAX_AMP) {
            if (ind_tmp + 1 == NLSF_MAX_AMP) {
                rate0_Q5 = rates_Q5[ind_tmp + NLSF_MAX_AMP];
                rate1_Q5 = 280;
            } else {
                rate0_Q5 = (280 - 43 * NLSF_MAX_AMP) + s_smulbb(43, ind_tmp);
                rate1_Q5 = rate0_Q5 + 43;
            }
        } else if (ind_tmp <= -NLSF_MAX_AMP) {
            if (ind_tmp == -NLSF_MAX_AMP) {
                rate0_Q5 = 280;
                rate1_Q5 = rates_Q5[ind_tmp + 1 + NLSF_MAX_AMP];
            } else {
                rate0_Q5 = (280 - 43 * NLSF_MAX_AMP) + s_smulbb(-43, ind_tmp);
                rate1_Q5 = rate0_Q5 - 43;
            }
        } else {
            rate0_Q5 = rates_Q5[ind_tmp + NLSF_MAX_AMP];
            rate1_Q5 = rates_Q5[ind_tmp + 1 + NLSF_MAX_AMP];
        }
        const i32 RD_tmp = RD[j];
        int diff_Q10 = in_Q10 - out0_Q10;
        RD[j] = s_addw(s_addw(RD_tmp, (i32)((u32)s_smulbb(diff_Q10, diff_Q10) * (u32)w_Q5i)), s_smulbb(mu_Q20, rate0_Q5));
        diff_Q10 = in_Q10 - out1_Q10;
        RD[j + NS] = s_addw(s_addw(RD_tmp, (i32)((u32)s_smulbb(diff_Q10, diff_Q10) * (u32)w_Q5i)), s_smulbb(mu_Q20, rate1_Q5));
    }
    if (NS <= (NLSF_DD_STATES >> 1)) {                                                      // the states double (:136-145)
#pragma unroll
        for (int j = 0; j < NS; j++) T.ind(j + NS, i) = (i8)(T.ind(j, i) + 1);
#pragma unroll
        for (int j = 2 * NS; j < NLSF_DD_STATES; j++) T.ind(j, i) = T.ind(j - 2 * NS, i);
    } else if (!last) {                                                                     // keep the best four of the eight (:146-199)
        i32 RD_min[NLSF_DD_STATES], RD_max[NLSF_DD_STATES], srt[NLSF_DD_STATES];
#pragma unroll
        for (int j = 0; j < NLSF_DD_STATES; j++) {
            if (RD[j] > RD[j + NLSF_DD_STATES]) {
                RD_max[j] = RD[j];
                RD_min[j] = RD[j + NLSF_DD_STATES];
                RD[j] = RD_min[j];
                RD[j + NLSF_DD_STATES] = RD_max[j];
                const i32 t = po[j];
                po[j] = po[j + NLSF_DD_STATES];
                po[j + NLSF_DD_STATES] = t;
                srt[j] = j + NLSF_DD_STATES;
            } else {
                RD_min[j] = RD[j];
                RD_max[j] = RD[j + NLSF_DD_STATES];
                srt[j] = j;
            }
        }
        while (1) {
            i32 min_max = 0x7FFFFFFF, max_min = 0;
            int ind_min_max = 0, ind_max_min = 0;
#pragma unroll
            for (int j = 0; j < NLSF_DD_STATES; j++) {
                if (min_max > RD_max[j]) { min_max = RD_max[j]; ind_min_max = j; }
                if (max_min < RD_min[j]) { max_min = RD_min[j]; ind_max_min = j; }
            }
            if (min_max >= max_min) break;
            nlsf_set4(srt, ind_max_min, nlsf_get4(srt, ind_min_max) ^ NLSF_DD_STATES);
            nlsf_set4(RD, ind_max_min, nlsf_get4(RD + NLSF_DD_STATES, ind_min_max));
            nlsf_set4(po, ind_max_min, nlsf_get4(po + NLSF_DD_STATES, ind_min_max));
            nlsf_set4(RD_min, ind_max_min, 0);
            nlsf_set4(RD_max, ind_min_max, 0x7FFFFFFF);
            for (int k = i; k < order; k++) T.ind(ind_max_min, k) = T.ind(ind_min_max, k);
        }
#pragma unroll
        for (int j = 0; j < NLSF_DD_STATES; j++) T.ind(j, i) = (i8)(T.ind(j, i) + (srt[j] >> NLSF_DD_STATES_LOG2));
    }
}

template <class TM>
CA_DEV i32 silk_NLSF_del_dec_quant_dev(TM &T, i8 *indices, const i16 *x_Q10, const i16 *w_Q5, const u8 *pred_coef_Q8, const i16 *ec_ix,
                                       const u8 *ec_rates_Q5, int quant_step_size_Q16, i32 inv_quant_step_size_Q6, i32 mu_Q20, int order)
{
    i32 RD[2 * NLSF_DD_STATES], po[2 * NLSF_DD_STATES];
#pragma unroll
    for (int j = 0; j < 2 * NLSF_DD_STATES; j++) { RD[j] = 0; po[j] = 0; }
    for (int i = order - 1; i >= 0; i--) {
        const u8 *rates_Q5 = &ec_rates_Q5[ec_ix[i]];
        const i32 pred_coef_Q16 = (i32)pred_coef_Q8[i] << 8;
        const int in_Q10 = x_Q10[i];
        const i32 w = (i32)w_Q5[i];
        // one state at the last coefficient, two at the one before, four from there on (NLSF_DD_STATES = 4; order >= 3)
        if (i == order - 1) nlsf_trellis_step<1>(T, RD, po, i, order, i == 0, in_Q10, w, pred_coef_Q16, rates_Q5, quant_step_size_Q16, inv_quant_step_size_Q6, mu_Q20);
        else if (i == order - 2) nlsf_trellis_step<2>(T, RD, po, i, order, i == 0, in_Q10, w, pred_coef_Q16, rates_Q5, quant_step_size_Q16, inv_quant_step_size_Q6, mu_Q20);
        else nlsf_trellis_step<4>(T, RD, po, i, order, i == 0, in_Q10, w, pred_coef_Q16, rates_Q5, quant_step_size_Q16, inv_quant_step_size_Q6, mu_Q20);
    }
    int ind_tmp = 0;
    i32 min_Q25 = 0x7FFFFFFF;
#pragma unroll
    for (int j = 0; j < 2 * NLSF_DD_STATES; j++) {
        if (min_Q25 > RD[j]) { min_Q25 = RD[j]; ind_tmp = j; }
    }
    for (int j = 0; j < order; j++) indices[j] = (i8)T.ind(ind_tmp & (NLSF_DD_STATES - 1), j);
    indices[0] = (i8)(indices[0] + (ind_tmp >> NLSF_DD_STATES_LOG2));
    return min_Q25;
}

CA_DEV void silk_NLSF_decode_dev(i16 *pNLSF_Q15, const i8 *NLSFIndices, const NlsfCB &cb)    // NLSF_decode.c:63-101
{
    u8 pred_Q8[SILK_MAX_LPC];
    i16 ec_ix[SILK_MAX_LPC], res_Q10[SILK_MAX_LPC], W_tmp_QW[SILK_MAX_LPC];
    const u8 *pCB_element = &cb.CB1_Q8[NLSFIndices[0] * cb.order];
    for (int i = 0; i < cb.order; i++) pNLSF_Q15[i] = (i16)((i32)pCB_element[i] << 7);
    silk_NLSF_unpack_dev(ec_ix, pred_Q8, cb, NLSFIndices[0]);
    {                                                                                       // silk_NLSF_residual_dequant, :35-58
        int out_Q10 = 0;
        for (int i = cb.order - 1; i >= 0; i--) {
            const int pred_Q10 = s_smulbb(out_Q10, (i32)pred_Q8[i]) >> 8;
            out_Q10 = shl32((i32)NLSFIndices[1 + i], 10);
            if (out_Q10 > 0) out_Q10 -= NLSF_LEVEL_ADJ_Q10;
            else if (out_Q10 < 0) out_Q10 += NLSF_LEVEL_ADJ_Q10;
            out_Q10 = s_smlawb(pred_Q10, out_Q10, cb.quantStepSize_Q16);
            res_Q10[i] = (i16)out_Q10;
        }
    }
    silk_NLSF_VQ_weights_laroia_dev(W_tmp_QW, pNLSF_Q15, cb.order);
    for (int i = 0; i < cb.order; i++) {
        const i32 W_tmp_Q9 = s_sqrt_approx((i32)W_tmp_QW[i] << (18 - NLSF_W_Q));
        const i32 t = (i32)pNLSF_Q15[i] + (((i32)res_Q10[i] << 14) / W_tmp_Q9);     // silk_DIV32_16 does not narrow its divisor
        pNLSF_Q15[i] = (i16)s_limit(t, 0, 32767);
    }
    silk_NLSF_stabilize_dev(pNLSF_Q15, cb.deltaMin_Q15, cb.order);
}

// silk_NLSF_encode (NLSF_encode.c:38-157): quantises pNLSF_Q15 in place, writes NLSFIndices[order + 1], returns the RD value
template <class TM>
CA_DEV i32 silk_NLSF_encode_dev(TM &T, i8 *NLSFIndices, i16 *pNLSF_Q15, const NlsfCB &cb, const i16 *pW_QW, int NLSF_mu_Q20, int nSurvivors,
                                int signalType)
{
    i32 err_Q26[NLSF_MAX_SURVIVORS];                     // nVectors <= 32
    int tempIndices1[NLSF_MAX_SURVIVORS];
    silk_NLSF_stabilize_dev(pNLSF_Q15, cb.deltaMin_Q15, cb.order);
    {                                                                                       // silk_NLSF_VQ, NLSF_VQ.c:35-68
        const u8 *p = cb.CB1_Q8;
        for (int i = 0; i < cb.nVectors; i++) {
            i32 sum_error_Q26 = 0;
            for (int m = 0; m < cb.order; m += 2) {
                i32 diff_Q15 = (i32)pNLSF_Q15[m] - ((i32)*p++ << 7);
                i32 sum_error_Q30 = s_smulbb(diff_Q15, diff_Q15);
                diff_Q15 = (i32)pNLSF_Q15[m + 1] - ((i32)*p++ << 7);
                sum_error_Q30 = s_addw(sum_error_Q30, s_smulbb(diff_Q15, diff_Q15));
                sum_error_Q26 = s_addw(sum_error_Q26, sum_error_Q30 >> 4);
            }
            err_Q26[i] = sum_error_Q26;
        }
    }
    silk_insertion_sort_increasing_dev(err_Q26, tempIndices1, cb.nVectors, nSurvivors);
    // survivors one after the other; only the best one's path is kept (the reference keeps all and picks the first minimum)
    i32 best_RD = 0;
    int best_s = -1;
    i8 best_path[SILK_MAX_LPC];
    for (int s = 0; s < nSurvivors; s++) {
        const int ind1 = tempIndices1[s];
        i16 res_Q10[SILK_MAX_LPC], NLSF_tmp_Q15[SILK_MAX_LPC], W_tmp_QW[SILK_MAX_LPC], W_adj_Q5[SILK_MAX_LPC], ec_ix[SILK_MAX_LPC];
        u8 pred_Q8[SILK_MAX_LPC];
        i8 path[SILK_MAX_LPC];
        const u8 *pCB_element = &cb.CB1_Q8[ind1 * cb.order];
#pragma unroll
        for (int i = 0; i < cb.order; i++) NLSF_tmp_Q15[i] = (i16)((i32)pCB_element[i] << 7);
        silk_NLSF_VQ_weights_laroia_dev(W_tmp_QW, NLSF_tmp_Q15, cb.order);
#pragma unroll
        for (int i = 0; i < cb.order; i++) {
            const i32 res_Q15 = (i16)(pNLSF_Q15[i] - NLSF_tmp_Q15[i]);
            const i32 W_tmp_Q9 = s_sqrt_approx((i32)W_tmp_QW[i] << (18 - NLSF_W_Q));
            res_Q10[i] = (i16)(s_smulbb(res_Q15, W_tmp_Q9) >> 14);
            W_adj_Q5[i] = (i16)(((i32)pW_QW[i] << 5) / (i32)W_tmp_QW[i]);
        }
        silk_NLSF_unpack_dev(ec_ix, pred_Q8, cb, ind1);
        i32 RD = silk_NLSF_del_dec_quant_dev(T, path, res_Q10, W_adj_Q5, pred_Q8, ec_ix, cb.ec_Rates_Q5, cb.quantStepSize_Q16,
                                             cb.invQuantStepSize_Q6, NLSF_mu_Q20, cb.order);
        const u8 *iCDF_ptr = &cb.CB1_iCDF[(signalType >> 1) * cb.nVectors];
        const int prob_Q8 = ind1 == 0 ? 256 - iCDF_ptr[ind1] : iCDF_ptr[ind1 - 1] - iCDF_ptr[ind1];
        const int bits_q7 = (8 << 7) - s_lin2log(prob_Q8);
        RD = s_addw(RD, s_smulbb(bits_q7, NLSF_mu_Q20 >> 2));
        if (best_s < 0 || RD < best_RD) {
            best_RD = RD;
            best_s = s;
            for (int i = 0; i < cb.order; i++) best_path[i] = path[i];
        }
    }
    NLSFIndices[0] = (i8)tempIndices1[best_s];
    for (int i = 0; i < cb.order; i++) NLSFIndices[1 + i] = best_path[i];
    silk_NLSF_decode_dev(pNLSF_Q15, NLSFIndices, cb);
    return best_RD;
}

// silk_process_NLSFs (process_NLSFs.c:35-106). pNLSF_Q15: in = silk_find_LPC_FIX's NLSFs, out = the quantised ones.
// T: the trellis' survivor state (NlsfTrellisLocal / NlsfTrellisCol above).
template <class TM>
CA_DEV void silk_process_NLSFs_order_dev(TM &T, i16 PredCoef_Q12[2][SILK_MAX_LPC], i8 *NLSFIndices, i16 *pNLSF_Q15, const i16 *prev_NLSFq_Q15,
                                         int speech_activity_Q8, int nb_subfr, const int order, int useInterpolatedNLSFs, int NLSFInterpCoef_Q2,
                                         int nSurvivors, int signalType, const NlsfTablesLds *tables)
{
    const NlsfCB cb = nlsf_codebook(order, tables);
    i16 pNLSF0_temp_Q15[SILK_MAX_LPC], pNLSFW_QW[SILK_MAX_LPC], pNLSFW0_temp_QW[SILK_MAX_LPC];
    // NLSF_mu = 0.003 - 0.001 * speech_activity  (SILK_FIX_CONST(0.003, 20) = 3146, SILK_FIX_CONST(-0.001, 28) = -268434)
    i32 NLSF_mu_Q20 = s_smlawb(3146, -268434, speech_activity_Q8);
    if (nb_subfr == 2) NLSF_mu_Q20 = NLSF_mu_Q20 + (NLSF_mu_Q20 >> 1);
    silk_NLSF_VQ_weights_laroia_dev(pNLSFW_QW, pNLSF_Q15, order);
    const bool doInterpolate = useInterpolatedNLSFs == 1 && NLSFInterpCoef_Q2 < 4;
    if (doInterpolate) {
        silk_interpolate_dev(pNLSF0_temp_Q15, prev_NLSFq_Q15, pNLSF_Q15, NLSFInterpCoef_Q2, order);
        silk_NLSF_VQ_weights_laroia_dev(pNLSFW0_temp_QW, pNLSF0_temp_Q15, order);
        const i32 i_sqr_Q15 = s_smulbb(NLSFInterpCoef_Q2, NLSFInterpCoef_Q2) << 11;
        for (int i = 0; i < order; i++)
            pNLSFW_QW[i] = (i16)s_smlawb(pNLSFW_QW[i] >> 1, (i32)pNLSFW0_temp_QW[i], i_sqr_Q15);
    }
    silk_NLSF_encode_dev(T, NLSFIndices, pNLSF_Q15, cb, pNLSFW_QW, NLSF_mu_Q20, nSurvivors, signalType);
    silk_NLSF2A_dev(PredCoef_Q12[1], pNLSF_Q15, order);
    if (doInterpolate) {
        silk_interpolate_dev(pNLSF0_temp_Q15, prev_NLSFq_Q15, pNLSF_Q15, NLSFInterpCoef_Q2, order);
        silk_NLSF2A_dev(PredCoef_Q12[0], pNLSF0_temp_Q15, order);
    } else {
        for (int i = 0; i < order; i++) PredCoef_Q12[0][i] = PredCoef_Q12[1][i];
    }
}

// The two LPC orders of the format (16 wideband, 10 narrow / medium band) as two instances of the inlined body: with the order a
// constant the per-coefficient loops unroll and the small per-survivor arrays they fill (residuals, weights, entropy-table
// offsets, predictor taps) are registers instead of run-time-indexed private memory.
template <class TM>
CA_DEV void silk_process_NLSFs_dev(TM &T, i16 PredCoef_Q12[2][SILK_MAX_LPC], i8 *NLSFIndices, i16 *pNLSF_Q15, const i16 *prev_NLSFq_Q15,
                                   int speech_activity_Q8, int nb_subfr, int order, int useInterpolatedNLSFs, int NLSFInterpCoef_Q2,
                                   int nSurvivors, int signalType, const NlsfTablesLds *tables = nullptr)
{
    if (order == 16)
        silk_process_NLSFs_order_dev(T, PredCoef_Q12, NLSFIndices, pNLSF_Q15, prev_NLSFq_Q15, speech_activity_Q8, nb_subfr, 16, useInterpolatedNLSFs,
                                     NLSFInterpCoef_Q2, nSurvivors, signalType, tables);
    else
        silk_process_NLSFs_order_dev(T, PredCoef_Q12, NLSFIndices, pNLSF_Q15, prev_NLSFq_Q15, speech_activity_Q8, nb_subfr, 10, useInterpolatedNLSFs,
                                     NLSFInterpCoef_Q2, nSurvivors, signalType, tables);
}

CA_DEV void silk_process_NLSFs_dev(i16 PredCoef_Q12[2][SILK_MAX_LPC], i8 *NLSFIndices, i16 *pNLSF_Q15, const i16 *prev_NLSFq_Q15,
                                   int speech_activity_Q8, int nb_subfr, int order, int useInterpolatedNLSFs, int NLSFInterpCoef_Q2,
                                   int nSurvivors, int signalType, const NlsfTablesLds *tables = nullptr)
{
    NlsfTrellisLocal T;
    silk_process_NLSFs_dev(T, PredCoef_Q12, NLSFIndices, pNLSF_Q15, prev_NLSFq_Q15, speech_activity_Q8, nb_subfr, order, useInterpolatedNLSFs,
                           NLSFInterpCoef_Q2, nSurvivors, signalType, tables);
}

// silk_residual_energy_FIX (residual_energy_FIX.c:37-98): x = LPC_in_pre, nb_subfr * (subfr_length + LPC_order) samples
template <class XA>
CA_DEV void silk_residual_energy_dev(i32 *nrgs, i32 *nrgsQ, XA x, const i16 a_Q12[2][SILK_MAX_LPC], const i32 *gains, int subfr_length,
                                     int nb_subfr, int LPC_order)
{
    const int offset = LPC_order + subfr_length;
    for (int i = 0; i < (nb_subfr >> 1); i++) {
        for (int j = 0; j < 2; j++) {
            int rshift;
            lpc_residual_energy(x, a_Q12[i], LPC_order, (2 * i + j) * offset + LPC_order, subfr_length, &nrgs[2 * i + j], &rshift);
            nrgsQ[2 * i + j] = -rshift;
        }
    }
    for (int i = 0; i < nb_subfr; i++) {
        const int lz1 = s_clz32(nrgs[i]) - 1, lz2 = s_clz32(gains[i]) - 1;
        i32 tmp32 = shl32(gains[i], lz2);
        tmp32 = s_smmul(tmp32, tmp32);
        nrgs[i] = s_smmul(tmp32, shl32(nrgs[i], lz1));
        nrgsQ[i] += lz1 + 2 * lz2 - 32 - 32;
    }
}

}  // namespace ca

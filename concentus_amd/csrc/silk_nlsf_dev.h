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
    i32 tmp1 = s_div_small(one, imax(NLSF_Q15[0], 1));
    i32 tmp2 = s_div_small(one, imax(NLSF_Q15[1] - NLSF_Q15[0], 1));
    W[0] = (i16)imin(tmp1 + tmp2, 32767);
#pragma unroll
    for (int k = 1; k < D - 1; k += 2) {
        tmp1 = s_div_small(one, imax(NLSF_Q15[k + 1] - NLSF_Q15[k], 1));
        W[k] = (i16)imin(tmp1 + tmp2, 32767);
        tmp2 = s_div_small(one, imax(NLSF_Q15[k + 2] - NLSF_Q15[k + 1], 1));
        W[k + 1] = (i16)imin(tmp1 + tmp2, 32767);
    }
    tmp1 = s_div_small(one, imax((1 << 15) - NLSF_Q15[D - 1], 1));
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

// ---- the delayed-decision trellis (NLSF_del_dec_quant.c:35-217) --------------------------------------------------------------
// R(v): what coding the index v costs with one entropy table, in Q5 bits -- the table's entry inside +-3, 280 at +-4, 43 more per
// step beyond. The reference's four-way branch (:100-128) evaluates exactly R(ind) and R(ind + 1); tabulated here for v = -10 .. 10
// per table, the trellis reads two entries and branches on nothing.
enum { NLSF_RATE_EXT = 2 * NLSF_MAX_AMP_EXT + 1, NLSF_EC_TABLES = 8 };
CA_DEV int nlsf_rate_ext(const u8 *rates_Q5 /* the nine entries of one table */, int v)
{
    const int a = v < 0 ? -v : v;
    return a < NLSF_MAX_AMP ? (int)rates_Q5[v + NLSF_MAX_AMP] : (280 - 43 * NLSF_MAX_AMP) + 43 * a;
}

// What silk_NLSF_encode recomputes for every survivor although it depends on the survivor's first-stage vector alone
// (NLSF_encode.c:93-103: the Laroia weights of the codebook vector and their square roots), and the extended rate tables.
// Same functions, evaluated once per workgroup (device: LDS) or once per process (host build).
struct NlsfEncTables {
    i16 wb_W_QW[32 * 16], nb_W_QW[32 * 10];
    u16 wb_W_Q9[32 * 16], nb_W_Q9[32 * 10];
    u16 wb_rate[NLSF_EC_TABLES * NLSF_RATE_EXT], nb_rate[NLSF_EC_TABLES * NLSF_RATE_EXT];
};

CA_DEV void nlsf_stage_enc_tables(NlsfEncTables &E, int tid, int nthreads)      // reads the constant tables: no ordering against nlsf_stage_tables
{
    for (int v = tid; v < 64; v += nthreads) {                                  // first-stage vectors: 0 .. 31 WB, 32 .. 63 NB / MB
        const bool wb = v < 32;
        const int vec = v & 31, order = wb ? 16 : 10;
        const u8 *cbv = wb ? &SILK_NLSF_WB_CB1_Q8[vec * 16] : &SILK_NLSF_NB_MB_CB1_Q8[vec * 10];
        i16 nlsf[SILK_MAX_LPC], W[SILK_MAX_LPC];
        for (int i = 0; i < order; i++) nlsf[i] = (i16)((i32)cbv[i] << 7);
        silk_NLSF_VQ_weights_laroia_dev(W, nlsf, order);
        for (int i = 0; i < order; i++) {
            (wb ? E.wb_W_QW : E.nb_W_QW)[vec * order + i] = W[i];
            (wb ? E.wb_W_Q9 : E.nb_W_Q9)[vec * order + i] = (u16)s_sqrt_approx((i32)W[i] << (18 - NLSF_W_Q));
        }
    }
    for (int e = tid; e < 2 * NLSF_EC_TABLES * NLSF_RATE_EXT; e += nthreads) {
        const bool wb = e < NLSF_EC_TABLES * NLSF_RATE_EXT;
        const int k = wb ? e : e - NLSF_EC_TABLES * NLSF_RATE_EXT, t = k / NLSF_RATE_EXT, v = k % NLSF_RATE_EXT - NLSF_MAX_AMP_EXT;
        (wb ? E.wb_rate : E.nb_rate)[k] = (u16)nlsf_rate_ext((wb ? SILK_NLSF_WB_ec_Rates_Q5 : SILK_NLSF_NB_MB_ec_Rates_Q5) + t * (2 * NLSF_MAX_AMP + 1), v);
    }
}
#if defined(CA_HOST_EMU)
inline const NlsfEncTables *nlsf_enc_tables_host()
{
    static const NlsfEncTables E = [] { NlsfEncTables e; nlsf_stage_enc_tables(e, 0, 1); return e; }();
    return &E;
}
#endif

// silk_NLSF_unpack for the trellis: the predictor taps and, instead of ec_ix (table * 9), the offset of the coefficient's
// extended rate table (table * 21)
template <int ORDER>
CA_DEV void nlsf_unpack_rates(i16 *rate_ix, u8 *pred_Q8, const NlsfCB &cb, int CB1_index)
{
    const u8 *ec_sel_ptr = &cb.ec_sel[CB1_index * ORDER / 2];
#pragma unroll
    for (int i = 0; i < ORDER; i += 2) {
        const int entry = *ec_sel_ptr++;
        rate_ix[i] = (i16)(((entry >> 1) & 7) * NLSF_RATE_EXT);
        pred_Q8[i] = cb.pred_Q8[i + (entry & 1) * (ORDER - 1)];
        rate_ix[i + 1] = (i16)(((entry >> 5) & 7) * NLSF_RATE_EXT);
        pred_Q8[i + 1] = cb.pred_Q8[i + ((entry >> 4) & 1) * (ORDER - 1) + 1];
    }
}

// element idx (run-time, 0 .. 3) of a four-entry register array: compare chains instead of memory
CA_DEV i32 nlsf_get4(const i32 *a, int idx) { return idx == 0 ? a[0] : idx == 1 ? a[1] : idx == 2 ? a[2] : a[3]; }
CA_DEV void nlsf_set4(i32 *a, int idx, i32 v) { a[0] = idx == 0 ? v : a[0]; a[1] = idx == 1 ? v : a[1]; a[2] = idx == 2 ? v : a[2]; a[3] = idx == 3 ? v : a[3]; }

// One coefficient of the trellis with NS (1, 2 or 4: a compile-time constant) live states. The reference keeps a row of indices
// per state and copies whole rows when a state is replaced (:183); here a state is, per coefficient, its index value and the
// state of the previous coefficient it continues (a trace-back pointer) -- a row copy moves exactly that pair --, four pairs
// packed into one word per coefficient (value + 16 in bits 0-4, pointer in bits 5-6 of each byte), the words in registers. The
// winner's row is read back through the pointers once the search is over (nlsf_traceback).
template <int NS>
CA_DEV u32 nlsf_trellis_step(i32 *RD /*[8]*/, i32 *po /*[8]*/, const bool last, const int in_Q10, const i32 w_Q5i, const i32 pred_coef_Q16,
                             const u16 *rext /* -> R(0) of the coefficient's table */, const int quant_step_size_Q16,
                             const i32 inv_quant_step_size_Q6, const i32 mu_Q20)
{
    i32 val[NLSF_DD_STATES], ptr[NLSF_DD_STATES];
#pragma unroll
    for (int j = 0; j < NS; j++) {
        const int pred_Q10 = s_smulwb(pred_coef_Q16, po[j]);
        const int res_Q10 = in_Q10 - pred_Q10;
        int ind_tmp = s_smulwb(inv_quant_step_size_Q6, res_Q10);
        ind_tmp = s_limit(ind_tmp, -NLSF_MAX_AMP_EXT, NLSF_MAX_AMP_EXT - 1);
        val[j] = ind_tmp;
        ptr[j] = j;
        // out0 / out1: the two reconstruction levels around the residual (the reference tabulates them per call, :61-79)
        int out0_Q10 = shl32(ind_tmp, 10), out1_Q10 = out0_Q10 + 1024;
        out0_Q10 += ind_tmp > 0 ? -NLSF_LEVEL_ADJ_Q10 : ind_tmp < 0 ? NLSF_LEVEL_ADJ_Q10 : 0;
        out1_Q10 += ind_tmp >= 0 ? -NLSF_LEVEL_ADJ_Q10 : ind_tmp < -1 ? NLSF_LEVEL_ADJ_Q10 : 0;
        out0_Q10 = s_smulwb(out0_Q10, quant_step_size_Q16) + pred_Q10;
        out1_Q10 = s_smulwb(out1_Q10, quant_step_size_Q16) + pred_Q10;
        po[j] = (i16)out0_Q10;
        po[j + NS] = (i16)out1_Q10;
        const int rate0_Q5 = rext[ind_tmp], rate1_Q5 = rext[ind_tmp + 1];
        const i32 RD_tmp = RD[j];
        int diff_Q10 = in_Q10 - out0_Q10;
        RD[j] = s_addw(s_addw(RD_tmp, (i32)((u32)s_smulbb(diff_Q10, diff_Q10) * (u32)w_Q5i)), s_smulbb(mu_Q20, rate0_Q5));
        diff_Q10 = in_Q10 - out1_Q10;
        RD[j + NS] = s_addw(s_addw(RD_tmp, (i32)((u32)s_smulbb(diff_Q10, diff_Q10) * (u32)w_Q5i)), s_smulbb(mu_Q20, rate1_Q5));
    }
    if (NS <= (NLSF_DD_STATES >> 1)) {                                                      // the states double (:136-145)
#pragma unroll
        for (int j = 0; j < NS; j++) { val[j + NS] = val[j] + 1; ptr[j + NS] = j; }
#pragma unroll
        for (int j = 2 * NS; j < NLSF_DD_STATES; j++) { val[j] = val[j - 2 * NS]; ptr[j] = ptr[j - 2 * NS]; }
    } else if (!last) {                                                                     // keep the best four of the eight (:146-199)
        i32 RD_min[NLSF_DD_STATES], RD_max[NLSF_DD_STATES], srt[NLSF_DD_STATES];
#pragma unroll
        for (int j = 0; j < NLSF_DD_STATES; j++) {
            const bool up = RD[j] > RD[j + NLSF_DD_STATES];
            const i32 lo = up ? RD[j + NLSF_DD_STATES] : RD[j], hi = up ? RD[j] : RD[j + NLSF_DD_STATES];
            const i32 pl = up ? po[j + NLSF_DD_STATES] : po[j], ph = up ? po[j] : po[j + NLSF_DD_STATES];
            RD_min[j] = lo; RD_max[j] = hi;
            RD[j] = lo; RD[j + NLSF_DD_STATES] = hi;
            po[j] = pl; po[j + NLSF_DD_STATES] = ph;
            srt[j] = up ? j + NLSF_DD_STATES : j;
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
            nlsf_set4(val, ind_max_min, nlsf_get4(val, ind_min_max));                        // the row copy (:183)
            nlsf_set4(ptr, ind_max_min, nlsf_get4(ptr, ind_min_max));
        }
#pragma unroll
        for (int j = 0; j < NLSF_DD_STATES; j++) val[j] += srt[j] >> NLSF_DD_STATES_LOG2;
    }
    u32 w = 0;
#pragma unroll
    for (int j = 0; j < NLSF_DD_STATES; j++) w |= (u32)((val[j] + 16) | (ptr[j] << 5)) << (8 * j);
    return w;
}

// The search over one survivor's residual: returns the best rate-distortion value, the trace-back words in tb[ORDER] and the
// winning candidate (0 .. 7: state | upper-candidate bit) in `win`.
template <int ORDER>
CA_DEV i32 nlsf_del_dec_quant(u32 *tb, int &win, const i16 *x_Q10, const i16 *w_Q5, const u8 *pred_coef_Q8, const i16 *rate_ix,
                              const u16 *rates_ext, const int quant_step_size_Q16, const i32 inv_quant_step_size_Q6, const i32 mu_Q20)
{
    i32 RD[2 * NLSF_DD_STATES], po[2 * NLSF_DD_STATES];
#pragma unroll
    for (int j = 0; j < 2 * NLSF_DD_STATES; j++) { RD[j] = 0; po[j] = 0; }
#pragma unroll
    for (int i = ORDER - 1; i >= 0; i--) {
        const u16 *rext = rates_ext + rate_ix[i] + NLSF_MAX_AMP_EXT;
        const i32 pred_coef_Q16 = (i32)pred_coef_Q8[i] << 8;
        const int in_Q10 = x_Q10[i];
        const i32 w = (i32)w_Q5[i];
        // one state at the last coefficient, two at the one before, four from there on (NLSF_DD_STATES = 4; ORDER >= 3)
        if (i == ORDER - 1) tb[i] = nlsf_trellis_step<1>(RD, po, i == 0, in_Q10, w, pred_coef_Q16, rext, quant_step_size_Q16, inv_quant_step_size_Q6, mu_Q20);
        else if (i == ORDER - 2) tb[i] = nlsf_trellis_step<2>(RD, po, i == 0, in_Q10, w, pred_coef_Q16, rext, quant_step_size_Q16, inv_quant_step_size_Q6, mu_Q20);
        else tb[i] = nlsf_trellis_step<4>(RD, po, i == 0, in_Q10, w, pred_coef_Q16, rext, quant_step_size_Q16, inv_quant_step_size_Q6, mu_Q20);
    }
    int ind_tmp = 0;
    i32 min_Q25 = 0x7FFFFFFF;
#pragma unroll
    for (int j = 0; j < 2 * NLSF_DD_STATES; j++) {
        if (min_Q25 > RD[j]) { min_Q25 = RD[j]; ind_tmp = j; }
    }
    win = ind_tmp;
    return min_Q25;
}

template <int ORDER>
CA_DEV void nlsf_traceback(i8 *indices, const u32 *tb, int win)                             // :207-213
{
    int s = win & (NLSF_DD_STATES - 1);
#pragma unroll
    for (int i = 0; i < ORDER; i++) {
        const u32 e = (tb[i] >> (8 * s)) & 0xff;
        indices[i] = (i8)((int)(e & 31) - 16);
        s = (int)(e >> 5);
    }
    indices[0] = (i8)(indices[0] + (win >> NLSF_DD_STATES_LOG2));
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

// silk_NLSF_encode (NLSF_encode.c:38-157): quantises pNLSF_Q15 in place, writes NLSFIndices[ORDER + 1], returns the RD value
template <int ORDER>
CA_DEV i32 silk_NLSF_encode_dev(i8 *NLSFIndices, i16 *pNLSF_Q15, const NlsfCB &cb, const NlsfEncTables &E, const i16 *pW_QW, int NLSF_mu_Q20,
                                int nSurvivors, int signalType)
{
    i32 err_Q26[NLSF_MAX_SURVIVORS];                     // nVectors <= 32
    int tempIndices1[NLSF_MAX_SURVIVORS];
    silk_NLSF_stabilize_dev(pNLSF_Q15, cb.deltaMin_Q15, ORDER);
    {                                                                                       // silk_NLSF_VQ, NLSF_VQ.c:35-68
        const u8 *p = cb.CB1_Q8;
        for (int i = 0; i < cb.nVectors; i++) {
            i32 sum_error_Q26 = 0;
#pragma unroll
            for (int m = 0; m < ORDER; m += 2) {
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
    const i16 *W_QW_tab = ORDER == 16 ? E.wb_W_QW : E.nb_W_QW;
    const u16 *W_Q9_tab = ORDER == 16 ? E.wb_W_Q9 : E.nb_W_Q9;
    const u16 *rates_ext = ORDER == 16 ? E.wb_rate : E.nb_rate;
    // survivors one after the other; only the best one's trace-back is kept (the reference keeps all paths and picks the first minimum)
    i32 best_RD = 0;
    int best_s = -1, best_win = 0;
    u32 best_tb[ORDER];
#pragma unroll
    for (int i = 0; i < ORDER; i++) best_tb[i] = 0;
    for (int s = 0; s < nSurvivors; s++) {
        const int ind1 = tempIndices1[s];
        i16 res_Q10[ORDER], W_adj_Q5[ORDER], rate_ix[ORDER];
        u8 pred_Q8[ORDER];
        const u8 *pCB_element = &cb.CB1_Q8[ind1 * ORDER];
        const i16 *W_tmp_QW = &W_QW_tab[ind1 * ORDER];
        const u16 *W_tmp_Q9 = &W_Q9_tab[ind1 * ORDER];
#pragma unroll
        for (int i = 0; i < ORDER; i++) {
            const i32 res_Q15 = (i16)(pNLSF_Q15[i] - (i16)((i32)pCB_element[i] << 7));
            res_Q10[i] = (i16)(s_smulbb(res_Q15, (i32)W_tmp_Q9[i]) >> 14);
            W_adj_Q5[i] = (i16)s_div_small((i32)pW_QW[i] << 5, (i32)W_tmp_QW[i]);
        }
        nlsf_unpack_rates<ORDER>(rate_ix, pred_Q8, cb, ind1);
        u32 tb[ORDER];
        int win;
        i32 RD = nlsf_del_dec_quant<ORDER>(tb, win, res_Q10, W_adj_Q5, pred_Q8, rate_ix, rates_ext, cb.quantStepSize_Q16, cb.invQuantStepSize_Q6,
                                           NLSF_mu_Q20);
        const u8 *iCDF_ptr = &cb.CB1_iCDF[(signalType >> 1) * cb.nVectors];
        const int prob_Q8 = ind1 == 0 ? 256 - iCDF_ptr[ind1] : iCDF_ptr[ind1 - 1] - iCDF_ptr[ind1];
        const int bits_q7 = (8 << 7) - s_lin2log(prob_Q8);
        RD = s_addw(RD, s_smulbb(bits_q7, NLSF_mu_Q20 >> 2));
        const bool better = best_s < 0 || RD < best_RD;
        best_RD = better ? RD : best_RD;
        best_s = better ? s : best_s;
        best_win = better ? win : best_win;
#pragma unroll
        for (int i = 0; i < ORDER; i++) best_tb[i] = better ? tb[i] : best_tb[i];
    }
    NLSFIndices[0] = (i8)tempIndices1[best_s];
    nlsf_traceback<ORDER>(NLSFIndices + 1, best_tb, best_win);
    silk_NLSF_decode_dev(pNLSF_Q15, NLSFIndices, cb);
    return best_RD;
}

// silk_process_NLSFs (process_NLSFs.c:35-106). pNLSF_Q15: in = silk_find_LPC_FIX's NLSFs, out = the quantised ones. The two LPC
// orders of the format (16 wideband, 10 narrow / medium band) are two instances of the body: with the order a constant the
// per-coefficient loops unroll and the small per-survivor arrays they fill (residuals, weights, rate-table offsets, predictor
// taps, the trace-back words) are registers instead of run-time-indexed private memory.
template <int ORDER>
CA_DEV void silk_process_NLSFs_order_dev(i16 PredCoef_Q12[2][SILK_MAX_LPC], i8 *NLSFIndices, i16 *pNLSF_Q15, const i16 *prev_NLSFq_Q15,
                                         int speech_activity_Q8, int nb_subfr, int useInterpolatedNLSFs, int NLSFInterpCoef_Q2,
                                         int nSurvivors, int signalType, const NlsfTablesLds *tables, const NlsfEncTables &E)
{
    const NlsfCB cb = nlsf_codebook(ORDER, tables);
    i16 pNLSF0_temp_Q15[SILK_MAX_LPC], pNLSFW_QW[SILK_MAX_LPC], pNLSFW0_temp_QW[SILK_MAX_LPC];
    // NLSF_mu = 0.003 - 0.001 * speech_activity  (SILK_FIX_CONST(0.003, 20) = 3146, SILK_FIX_CONST(-0.001, 28) = -268434)
    i32 NLSF_mu_Q20 = s_smlawb(3146, -268434, speech_activity_Q8);
    if (nb_subfr == 2) NLSF_mu_Q20 = NLSF_mu_Q20 + (NLSF_mu_Q20 >> 1);
    silk_NLSF_VQ_weights_laroia_dev(pNLSFW_QW, pNLSF_Q15, ORDER);
    const bool doInterpolate = useInterpolatedNLSFs == 1 && NLSFInterpCoef_Q2 < 4;
    if (doInterpolate) {
        silk_interpolate_dev(pNLSF0_temp_Q15, prev_NLSFq_Q15, pNLSF_Q15, NLSFInterpCoef_Q2, ORDER);
        silk_NLSF_VQ_weights_laroia_dev(pNLSFW0_temp_QW, pNLSF0_temp_Q15, ORDER);
        const i32 i_sqr_Q15 = s_smulbb(NLSFInterpCoef_Q2, NLSFInterpCoef_Q2) << 11;
        for (int i = 0; i < ORDER; i++)
            pNLSFW_QW[i] = (i16)s_smlawb(pNLSFW_QW[i] >> 1, (i32)pNLSFW0_temp_QW[i], i_sqr_Q15);
    }
    silk_NLSF_encode_dev<ORDER>(NLSFIndices, pNLSF_Q15, cb, E, pNLSFW_QW, NLSF_mu_Q20, nSurvivors, signalType);
    silk_NLSF2A_dev(PredCoef_Q12[1], pNLSF_Q15, ORDER);
    if (doInterpolate) {
        silk_interpolate_dev(pNLSF0_temp_Q15, prev_NLSFq_Q15, pNLSF_Q15, NLSFInterpCoef_Q2, ORDER);
        silk_NLSF2A_dev(PredCoef_Q12[0], pNLSF0_temp_Q15, ORDER);
    } else {
        for (int i = 0; i < ORDER; i++) PredCoef_Q12[0][i] = PredCoef_Q12[1][i];
    }
}

// tables / enc: the workgroup's LDS copies in the kernels (nlsf_stage_tables, nlsf_stage_enc_tables); the host build may pass
// neither (constant tables; the derived ones built on first use)
CA_DEV void silk_process_NLSFs_dev(i16 PredCoef_Q12[2][SILK_MAX_LPC], i8 *NLSFIndices, i16 *pNLSF_Q15, const i16 *prev_NLSFq_Q15,
                                   int speech_activity_Q8, int nb_subfr, int order, int useInterpolatedNLSFs, int NLSFInterpCoef_Q2,
                                   int nSurvivors, int signalType, const NlsfTablesLds *tables = nullptr, const NlsfEncTables *enc = nullptr)
{
#if defined(CA_HOST_EMU)
    if (!enc) enc = nlsf_enc_tables_host();
#endif
    if (order == 16)
        silk_process_NLSFs_order_dev<16>(PredCoef_Q12, NLSFIndices, pNLSF_Q15, prev_NLSFq_Q15, speech_activity_Q8, nb_subfr, useInterpolatedNLSFs,
                                         NLSFInterpCoef_Q2, nSurvivors, signalType, tables, *enc);
    else
        silk_process_NLSFs_order_dev<10>(PredCoef_Q12, NLSFIndices, pNLSF_Q15, prev_NLSFq_Q15, speech_activity_Q8, nb_subfr, useInterpolatedNLSFs,
                                         NLSFInterpCoef_Q2, nSurvivors, signalType, tables, *enc);
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

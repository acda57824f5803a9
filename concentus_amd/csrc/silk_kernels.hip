// silk_kernels.hip -- the two SILK fixed-point inner loops named by the north star, batched over
// function-boundary records (BASELINE config #4).
//
//   silk_burg_kernel  <- silk_burg_modified_c        opus-fix/silk/fixed/burg_modified_FIX.c:45-275
//   silk_nsq_kernel   <- silk_NSQ_c                  opus-fix/silk/NSQ.c:74-180
//                        silk_noise_shape_quantizer  opus-fix/silk/NSQ.c:183-421
//                        silk_nsq_scale_states       opus-fix/silk/NSQ.c:423-496
//                        silk_LPC_analysis_filter    opus-fix/silk/LPC_analysis_filter.c:47-108 (FIXED_POINT branch)
//
// Mapping: both loops are recurrences that are serial in their own index (Burg: order n depends on
// order n-1; NSQ: sample i feeds back into sample i+1 through four filters), with only ~16-wide inner
// products. A wavefront per record would leave 48 lanes idle and pay a cross-lane reduction per
// sample, so here ONE LANE owns ONE record and a wavefront advances 64 independent records in
// lock-step: the loop trip counts (order, subframe length) are the same for every record of a batch, so
// the wave stays converged, per-lane temporaries are indexed uniformly (scratch accesses coalesce), the
// short filter states (sAR2, the last 16 sLPC samples) live in registers, and the only divergent
// addresses are the pitch-lag taps.
//
// Arithmetic: the reference's x86-64 build uses the 64-bit macro forms (OPUS_FAST_INT64, silk/macros.h:47-102);
// 16x32 products are evaluated on split halves (full-rate 24-bit multiplier), 32x32 ones as 64-bit.
#include <string.h>
#include "silk_burg_dev.h"
#include "opusgpu_internal.h"
#include "../../include/opusgpu_silk.h"
#include "silk_validate.h"

namespace ca {


// ---- silk_burg_modified: one lane per record --------------------------------------------------------
// A record's samples x[384] are read ~25 times each (16 autocorrelation lags, then both ends of every subframe in
// every order of the recursion). From HBM that is a 2-byte load per use at a 784-byte stride between lanes -- one cache
// line per lane per instruction. The wavefront instead copies its 64 records' samples into LDS once (16-byte loads),
// laid out [sample][lane], and every later use is a conflict-free LDS read.
struct BurgX {
    const i16 *p;                                              // this lane's column of the [sample][lane] block
    __device__ __forceinline__ i32 operator[](int k) const { return p[k * 64]; }
    __device__ __forceinline__ BurgX operator+(int o) const { BurgX r; r.p = p + o * 64; return r; }
};

__global__ __launch_bounds__(64) void silk_burg_kernel(const opusgpu_burg_in *__restrict__ recs, opusgpu_burg_out *__restrict__ outs, int n_rec,
                                                       int *__restrict__ bad_records)
{
    __shared__ i16 edge_s[BURG_EDGE_SLOTS * 64];               // the recursion's subframe edges, [slot][lane]: 16 KB, four workgroups per CU
    i16 xs[OPUSGPU_SILK_BURG_MAX_X];                           // private: the energy and lag-product passes read it once each, in order
    const int r = blockIdx.x * 64 + threadIdx.x;
    if (r >= n_rec) return;
    const opusgpu_burg_in &in = recs[r];
    if (!burg_record_ok(in)) {                                 // would index xs[] / the order-16 arrays out of bounds
        opusgpu_burg_out &o = outs[r];
        o.res_nrg = 0;
        o.res_nrg_Q = (i32)0x80000000;
        for (int k = 0; k < 16; k++) o.A_Q16[k] = 0;
        atomicAdd(bad_records, 1);
        return;
    }
    {
        static_assert(sizeof(opusgpu_burg_in) % 16 == 0 && OPUSGPU_SILK_BURG_MAX_X % 8 == 0, "16-byte loads of x");
        const int nx = in.subfr_length * in.nb_subfr;
        const int4 *src = reinterpret_cast<const int4 *>(in.x);
        for (int k = 0; k < nx; k += 8) {
            const int4 w = src[k >> 3];
            xs[k + 0] = (i16)w.x; xs[k + 1] = (i16)(w.x >> 16); xs[k + 2] = (i16)w.y; xs[k + 3] = (i16)(w.y >> 16);
            xs[k + 4] = (i16)w.z; xs[k + 5] = (i16)(w.z >> 16); xs[k + 6] = (i16)w.w; xs[k + 7] = (i16)(w.w >> 16);
        }
    }
    BurgEdgesCol e;
    e.p = edge_s + threadIdx.x;
    e.stage((const i16 *)xs, in.subfr_length, in.nb_subfr);
    silk_burg_modified_dev((const i16 *)xs, e, in.minInvGain_Q30, in.subfr_length, in.nb_subfr, in.D, outs[r].A_Q16, &outs[r].res_nrg, &outs[r].res_nrg_Q);
}

// ---- silk_NSQ: one lane per record ---------------------------------------------------------------------
// ws: per-record scratch in HBM: the scaled re-whitened prediction buffer int32 sLTP_Q15[640] (the int16 sLTP[640] behind it is
// unused since the re-whitening writes its outputs scaled; it stays in the workspace's size).
struct NsqScratch { i32 sLTP_Q15[640]; i16 sLTP[640]; };

__global__ __launch_bounds__(64) void silk_nsq_kernel(const opusgpu_nsq_in *__restrict__ recs, opusgpu_nsq_state *__restrict__ states,
                                                      opusgpu_nsq_out *__restrict__ outs, NsqScratch *__restrict__ ws, int n_rec,
                                                      int *__restrict__ bad_records, const int *__restrict__ rows)
{
    int r = blockIdx.x * 64 + threadIdx.x;
    if (r >= n_rec) return;
    if (rows) r = rows[r];                                     // the bitrate loop's second passes: a list of frames, in place
    const opusgpu_nsq_in &in = recs[r];
    opusgpu_nsq_state &NSQ = states[r];
    if (!nsq_record_ok(in, NSQ.lagPrev)) {                     // state untouched, no pulses
        for (int k = 0; k < OPUSGPU_SILK_MAX_FRAME; k++) outs[r].pulses[k] = 0;
        atomicAdd(bad_records, 1);
        return;
    }
    i32 *sLTP_Q15 = ws[r].sLTP_Q15;
    const int nb_subfr = in.nb_subfr, subfr_length = in.subfr_length, frame_length = in.frame_length;
    const int ltp_mem_length = in.ltp_mem_length, predictLPCOrder = in.predictLPCOrder, shapingLPCOrder = in.shapingLPCOrder;
    const int signalType = in.signalType;
    const int voiced = signalType == 2;
    i8 *pulses = outs[r].pulses;
    i32 rand_seed = in.Seed;
    int lag = NSQ.lagPrev;
    const int offset_Q10 = voiced ? (in.quantOffsetType ? 100 : 32) : (in.quantOffsetType ? 240 : 100);   // silk/tables_other.c:95-97
    const int LSF_interpolation_flag = in.NLSFInterpCoef_Q2 == 4 ? 0 : 1;
    int sLTP_shp_buf_idx = ltp_mem_length, sLTP_buf_idx = ltp_mem_length;
    i32 prev_gain_Q16 = NSQ.prev_gain_Q16;
    i32 sLF_AR_shp_Q14 = NSQ.sLF_AR_shp_Q14;
    int rewhite_flag = 0;
    // short filter states in registers: sAR2_Q14[16] and the last 16 sLPC_Q14 samples (lp[0] = newest)
    i32 ar[16], lp[16];
#pragma unroll
    for (int j = 0; j < 16; j++) { ar[j] = NSQ.sAR2_Q14[j]; lp[j] = NSQ.sLPC_Q14[31 - j]; }
    i32 lpc_old[16];                  // sLPC_Q14[0..16): the older half of the 32-sample history (state output only)
#pragma unroll
    for (int j = 0; j < 16; j++) lpc_old[j] = NSQ.sLPC_Q14[j];

    for (int k = 0; k < nb_subfr; k++) {
        const i16 *A_Q12 = &in.PredCoef_Q12[((k >> 1) | (1 - LSF_interpolation_flag)) * 16];
        const i16 *B_Q14 = &in.LTPCoef_Q14[k * 5];
        const i16 *AR_shp_Q13 = &in.AR2_Q13[k * 16];
        i32 HarmShapeFIRPacked_Q14 = in.HarmShapeGain_Q14[k] >> 2;
        HarmShapeFIRPacked_Q14 |= shl32(in.HarmShapeGain_Q14[k] >> 1, 16);
        rewhite_flag = 0;
        if (voiced) {
            lag = in.pitchL[k];
            if ((k & (3 - (LSF_interpolation_flag << 1))) == 0) {
                const int start_idx = ltp_mem_length - lag - predictLPCOrder - 5 / 2;
                // silk_LPC_analysis_filter (celt_fir form): out = SAT16(in + PSHR32(-sum B[m] in[ix-1-m], 12)), fused with the scaling
                // silk_nsq_scale_states applies to exactly these lag + 2 outputs (NSQ.c:445-456: sLTP_Q15[i] = SMULWB(inv_gain_Q31,
                // sLTP[i])): the 16-bit sLTP buffer in between is never written or read, four results leave per 16-byte store
                const i16 *inp = &NSQ.xq[start_idx + k * subfr_length];
                i32 *outq = &sLTP_Q15[start_idx];
                const int len = ltp_mem_length - start_idx;
                i32 ig_Q31 = s_inverse32_varq(in.Gains_Q16[k] > 1 ? in.Gains_Q16[k] : 1, 47);
                if (k == 0) ig_Q31 = shl32(s_smulwb(ig_Q31, in.LTP_scale_Q14), 2);
                // the predictLPCOrder previous input samples travel in a register window: every sample of xq is read once
                i32 nA[16], w[16];
#pragma unroll
                for (int m = 0; m < 16; m++) {
                    nA[m] = m < predictLPCOrder ? (i32)(i16)(-A_Q12[m]) : 0;
                    w[m] = m < predictLPCOrder ? (i32)inp[predictLPCOrder - 1 - m] : 0;
                }
                struct __attribute__((packed, aligned(2))) H4 { i16 v[4]; };
                struct __attribute__((packed, aligned(4))) W4 { i32 v[4]; };
                int ix = predictLPCOrder;
                for (; ix + 4 <= len; ix += 4) {
                    const H4 xi4 = *reinterpret_cast<const H4 *>(&inp[ix]);
                    W4 o;
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        i32 sum = 0;
#pragma unroll
                        for (int m = 0; m < 16; m++) sum = s_addw(sum, __mul24(nA[m], w[m]));
                        const i32 xi = (i32)xi4.v[u];
                        i32 v = xi + pshr32(sum, 12);
                        v = v > 32767 ? 32767 : (v < -32768 ? -32768 : v);
                        o.v[u] = s_smulwb(ig_Q31, v);
#pragma unroll
                        for (int m = 15; m > 0; m--) w[m] = w[m - 1];
                        w[0] = xi;
                    }
                    *reinterpret_cast<W4 *>(&outq[ix]) = o;
                }
                for (; ix < len; ix++) {
                    i32 sum = 0;
#pragma unroll
                    for (int m = 0; m < 16; m++) sum = s_addw(sum, __mul24(nA[m], w[m]));
                    const i32 xi = (i32)inp[ix];
                    i32 v = xi + pshr32(sum, 12);
                    v = v > 32767 ? 32767 : (v < -32768 ? -32768 : v);
                    outq[ix] = s_smulwb(ig_Q31, v);
#pragma unroll
                    for (int m = 15; m > 0; m--) w[m] = w[m - 1];
                    w[0] = xi;
                }
                rewhite_flag = 1;
                sLTP_buf_idx = ltp_mem_length;
            }
        }
        // ---- silk_nsq_scale_states (NSQ.c:423-496) ----
        const i32 gain = in.Gains_Q16[k];
        const i32 inv_gain_Q31 = s_inverse32_varq(gain > 1 ? gain : 1, 47);
        const i32 gain_adj_Q16 = gain != prev_gain_Q16 ? s_div32_varq(prev_gain_Q16, gain, 16) : (i32)1 << 16;
        const i32 inv_gain_Q23 = s_rshift_round(inv_gain_Q31, 8);
        prev_gain_Q16 = gain;
        {
            const int lg = in.pitchL[k];
            // (rewhite_flag: sLTP_Q15[idx - lag - 2 .. idx) was scaled where it was produced, above)
            if (gain_adj_Q16 != (i32)1 << 16) {
                // Four values per access (the record is 4-byte aligned), and the loads of eight accesses in flight before the first
                // store: written as load - scale - store per element the loop is one exposed memory round trip per trip.
                struct __attribute__((packed, aligned(4))) Q4 { i32 v[4]; };
                auto scale_run = [&](i32 *a, int i, const int end) {
                    for (; i + 32 <= end; i += 32) {
                        Q4 q[8];
#pragma unroll
                        for (int c = 0; c < 8; c++) q[c] = *reinterpret_cast<const Q4 *>(&a[i + 4 * c]);
#pragma unroll
                        for (int c = 0; c < 8; c++) {
#pragma unroll
                            for (int u = 0; u < 4; u++) q[c].v[u] = s_smulww(gain_adj_Q16, q[c].v[u]);
                            *reinterpret_cast<Q4 *>(&a[i + 4 * c]) = q[c];
                        }
                    }
                    for (; i + 4 <= end; i += 4) {
                        Q4 q = *reinterpret_cast<const Q4 *>(&a[i]);
#pragma unroll
                        for (int u = 0; u < 4; u++) q.v[u] = s_smulww(gain_adj_Q16, q.v[u]);
                        *reinterpret_cast<Q4 *>(&a[i]) = q;
                    }
                    for (; i < end; i++) a[i] = s_smulww(gain_adj_Q16, a[i]);
                };
                scale_run(NSQ.sLTP_shp_Q14, sLTP_shp_buf_idx - ltp_mem_length, sLTP_shp_buf_idx);
                if (voiced && rewhite_flag == 0) scale_run(sLTP_Q15, sLTP_buf_idx - lg - 5 / 2, sLTP_buf_idx);
                sLF_AR_shp_Q14 = s_smulww(gain_adj_Q16, sLF_AR_shp_Q14);
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    lp[j] = s_smulww(gain_adj_Q16, lp[j]);
                    lpc_old[j] = s_smulww(gain_adj_Q16, lpc_old[j]);
                    ar[j] = s_smulww(gain_adj_Q16, ar[j]);
                }
            }
        }
        // ---- silk_noise_shape_quantizer (NSQ.c:183-421) ----
        const i32 Gain_Q10 = gain >> 6;
        const int Tilt_Q14 = in.Tilt_Q14[k], Lambda_Q10 = in.Lambda_Q10;
        const i32 LF_shp_Q14 = in.LF_shp_Q14[k];
        int shp_lag = sLTP_shp_buf_idx - lag + 3 / 2;          // index of shp_lag_ptr[0]
        int pred_lag = sLTP_buf_idx - lag + 5 / 2;             // index of pred_lag_ptr[0]
        i32 shp_prev = NSQ.sLTP_shp_Q14[sLTP_shp_buf_idx - 1]; // sLTP_shp_Q14[idx-1], carried in a register
        i32 a12[16], ar13[16];
#pragma unroll
        for (int j = 0; j < 16; j++) { a12[j] = A_Q12[j]; ar13[j] = AR_shp_Q13[j]; }
        const i32 b0 = B_Q14[0], b1 = B_Q14[1], b2 = B_Q14[2], b3 = B_Q14[3], b4 = B_Q14[4];
        const i32 *x_Q3 = in.x_Q3 + k * subfr_length;
        i16 *pxq = &NSQ.xq[ltp_mem_length + k * subfr_length];
        // The pitch-lag taps slide by one sample per step: they are kept in registers (pt[j] = sLTP_Q15[pred_lag - j],
        // st[j] = sLTP_shp_Q14[shp_lag - j]) and only the newest one is loaded. A step writes position idx and the newest
        // taps of the NEXT step sit at idx + 3 - lag / idx + 2 - lag, i.e. were written at least a step ago when lag >= 4
        // (pitch lags are >= 2 ms): they are fetched a whole step ahead, like the next input sample.
        i32 pt[5] = {0, 0, 0, 0, 0}, st[3] = {0, 0, 0}, pn = 0, sn = 0;
        if (voiced) {
#pragma unroll
            for (int j = 0; j < 5; j++) pt[j] = sLTP_Q15[pred_lag - j];
        }
        if (lag > 0) {
#pragma unroll
            for (int j = 0; j < 3; j++) st[j] = NSQ.sLTP_shp_Q14[shp_lag - j];
        }
        struct __attribute__((packed, aligned(4))) Q4 { i32 v[4]; };
        // ---- the usual case (no pitch lag, or one of >= 12 samples; subframes of 4n samples), four samples per group ----
        // What limits the kernel is the memory pipeline, not arithmetic: a lane's load or store of its own record occupies it for a
        // cache line whatever its width, and loads queue behind earlier stores (one counter orders both). So a group's three inputs
        // (x_Q3, the prediction tap, the shaping tap: four values each) are ONE 16-byte load each, issued BEFORE the previous group's
        // five 16-byte stores, whose completion then overlaps the next group's arithmetic. Inside a group the two 16-deep
        // histories (sLPC_Q14's newest samples, the sAR2 delay line) are not shifted per sample: sample u reads them through
        // compile-time indices into [the group's new values | the registers as they were at the group's start].
        const bool fast = (lag <= 0 || lag >= 12) && (subfr_length & 3) == 0;
        if (fast) {
            i32 ar13z[16];
#pragma unroll
            for (int j = 0; j < 16; j++) ar13z[j] = j < shapingLPCOrder ? ar13[j] : 0;       // the delay line beyond the order adds nothing
            Q4 xg, pg, sg;
#pragma unroll
            for (int u = 0; u < 4; u++) { pg.v[u] = 0; sg.v[u] = 0; }
            xg = *reinterpret_cast<const Q4 *>(&x_Q3[0]);
            if (voiced) pg = *reinterpret_cast<const Q4 *>(&sLTP_Q15[pred_lag + 1]);
            if (lag > 0) sg = *reinterpret_cast<const Q4 *>(&NSQ.sLTP_shp_Q14[shp_lag + 1]);
            for (int i0 = 0; i0 < subfr_length; i0 += 4) {
                i32 nw[4] = {0, 0, 0, 0}, sv[4] = {0, 0, 0, 0}, o_shp[4], o_ltp[4];
                u32 o_xq[4], o_pl[4];
                const int shp_idx0 = sLTP_shp_buf_idx, ltp_idx0 = sLTP_buf_idx;
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    rand_seed = (i32)(907633515u + (u32)rand_seed * 196314165u);
                    sv[u] = u == 0 ? lp[0] : nw[u - 1];                                    // the newest sLPC_Q14 sample enters the delay line
                    i32 LPC_pred_Q10 = predictLPCOrder >> 1;
#pragma unroll
                    for (int j = 0; j < 10; j++) LPC_pred_Q10 = s_smlawb(LPC_pred_Q10, j < u ? nw[u - 1 - j] : lp[j - u], a12[j]);
                    if (predictLPCOrder == 16) {
#pragma unroll
                        for (int j = 10; j < 16; j++) LPC_pred_Q10 = s_smlawb(LPC_pred_Q10, j < u ? nw[u - 1 - j] : lp[j - u], a12[j]);
                    }
                    i32 LTP_pred_Q13 = 0;
                    if (voiced) {
                        LTP_pred_Q13 = 2;
                        LTP_pred_Q13 = s_smlawb(LTP_pred_Q13, pt[0], b0);
                        LTP_pred_Q13 = s_smlawb(LTP_pred_Q13, pt[1], b1);
                        LTP_pred_Q13 = s_smlawb(LTP_pred_Q13, pt[2], b2);
                        LTP_pred_Q13 = s_smlawb(LTP_pred_Q13, pt[3], b3);
                        LTP_pred_Q13 = s_smlawb(LTP_pred_Q13, pt[4], b4);
                    }
                    // noise shape feedback (NSQ.c:262-279): after the shift the delay line holds [sv[u], sv[u-1], .., ar[0], ar[1], ..]
                    i32 n_AR_Q12 = shapingLPCOrder >> 1;
#pragma unroll
                    for (int j = 0; j < 16; j++) n_AR_Q12 = s_smlawb(n_AR_Q12, j <= u ? sv[u - j] : ar[j - u - 1], ar13z[j]);
                    n_AR_Q12 = shl32(n_AR_Q12, 1);
                    n_AR_Q12 = s_smlawb(n_AR_Q12, sLF_AR_shp_Q14, Tilt_Q14);
                    i32 n_LF_Q12 = s_smulwb(shp_prev, LF_shp_Q14);
                    n_LF_Q12 = s_smlawt(n_LF_Q12, sLF_AR_shp_Q14, LF_shp_Q14);
                    i32 tmp1 = s_subw(shl32(LPC_pred_Q10, 2), n_AR_Q12);
                    tmp1 = s_subw(tmp1, n_LF_Q12);
                    if (lag > 0) {
                        i32 n_LTP_Q13 = s_smulwb(s_addw(st[0], st[2]), HarmShapeFIRPacked_Q14);
                        n_LTP_Q13 = s_smlawt(n_LTP_Q13, st[1], HarmShapeFIRPacked_Q14);
                        n_LTP_Q13 = shl32(n_LTP_Q13, 1);
                        const i32 tmp2 = s_subw(LTP_pred_Q13, n_LTP_Q13);
                        tmp1 = s_addw(tmp2, shl32(tmp1, 1));
                        tmp1 = s_rshift_round(tmp1, 3);
                    } else {
                        tmp1 = s_rshift_round(tmp1, 2);
                    }
                    i32 r_Q10 = s_subw(s_smulww(xg.v[u], inv_gain_Q23), tmp1);
                    if (rand_seed < 0) r_Q10 = (i32)(0u - (u32)r_Q10);
                    r_Q10 = s_limit(r_Q10, -(31 << 10), 30 << 10);
                    i32 q1_Q10 = r_Q10 - offset_Q10, q2_Q10, rd1_Q20, rd2_Q20;
                    const i32 q1_Q0 = q1_Q10 >> 10;
                    if (q1_Q0 > 0) {
                        q1_Q10 = shl32(q1_Q0, 10) - 80 + offset_Q10;
                        q2_Q10 = q1_Q10 + 1024;
                        rd1_Q20 = s_smulbb(q1_Q10, Lambda_Q10);
                        rd2_Q20 = s_smulbb(q2_Q10, Lambda_Q10);
                    } else if (q1_Q0 == 0) {
                        q1_Q10 = offset_Q10;
                        q2_Q10 = q1_Q10 + (1024 - 80);
                        rd1_Q20 = s_smulbb(q1_Q10, Lambda_Q10);
                        rd2_Q20 = s_smulbb(q2_Q10, Lambda_Q10);
                    } else if (q1_Q0 == -1) {
                        q2_Q10 = offset_Q10;
                        q1_Q10 = q2_Q10 - (1024 - 80);
                        rd1_Q20 = s_smulbb(-q1_Q10, Lambda_Q10);
                        rd2_Q20 = s_smulbb(q2_Q10, Lambda_Q10);
                    } else {
                        q1_Q10 = shl32(q1_Q0, 10) + 80 + offset_Q10;
                        q2_Q10 = q1_Q10 + 1024;
                        rd1_Q20 = s_smulbb(-q1_Q10, Lambda_Q10);
                        rd2_Q20 = s_smulbb(-q2_Q10, Lambda_Q10);
                    }
                    i32 rr_Q10 = r_Q10 - q1_Q10;
                    rd1_Q20 = s_addw(rd1_Q20, s_smulbb(rr_Q10, rr_Q10));
                    rr_Q10 = r_Q10 - q2_Q10;
                    rd2_Q20 = s_addw(rd2_Q20, s_smulbb(rr_Q10, rr_Q10));
                    if (rd2_Q20 < rd1_Q20) q1_Q10 = q2_Q10;
                    const i32 pulse = (i8)s_rshift_round(q1_Q10, 10);
                    o_pl[u] = (u32)pulse & 0xffu;
                    i32 exc_Q14 = shl32(q1_Q10, 4);
                    if (rand_seed < 0) exc_Q14 = (i32)(0u - (u32)exc_Q14);
                    const i32 LPC_exc_Q14 = s_addw(exc_Q14, shl32(LTP_pred_Q13, 1));
                    const i32 xq_Q14 = s_addw(LPC_exc_Q14, shl32(LPC_pred_Q10, 4));
                    {   // silk_SAT16(silk_RSHIFT_ROUND(silk_SMULWW(xq_Q14, Gain_Q10), 8)) in 64 bits
                        i64 t = ((i64)xq_Q14 * Gain_Q10) >> 16;
                        t = ((t >> 7) + 1) >> 1;
                        o_xq[u] = (u32)(i32)(t > 32767 ? 32767 : (t < -32768 ? -32768 : t)) & 0xffffu;
                    }
                    nw[u] = xq_Q14;
                    sLF_AR_shp_Q14 = s_subw(xq_Q14, shl32(n_AR_Q12, 2));
                    shp_prev = s_subw(sLF_AR_shp_Q14, shl32(n_LF_Q12, 2));
                    o_shp[u] = shp_prev;
                    o_ltp[u] = shl32(LPC_exc_Q14, 1);
                    rand_seed = (i32)((u32)rand_seed + (u32)pulse);
                    pt[4] = pt[3]; pt[3] = pt[2]; pt[2] = pt[1]; pt[1] = pt[0]; pt[0] = pg.v[u];
                    st[2] = st[1]; st[1] = st[0]; st[0] = sg.v[u];
                }
                sLTP_shp_buf_idx += 4;
                sLTP_buf_idx += 4;
                pred_lag += 4;
                shp_lag += 4;
                if (i0 + 4 < subfr_length) {                                                // the next group's inputs, ahead of this group's stores
                    xg = *reinterpret_cast<const Q4 *>(&x_Q3[i0 + 4]);
                    if (voiced) pg = *reinterpret_cast<const Q4 *>(&sLTP_Q15[pred_lag + 1]);
                    if (lag > 0) sg = *reinterpret_cast<const Q4 *>(&NSQ.sLTP_shp_Q14[shp_lag + 1]);
                }
                *reinterpret_cast<u32 *>(&pulses[k * subfr_length + i0]) = o_pl[0] | (o_pl[1] << 8) | (o_pl[2] << 16) | (o_pl[3] << 24);
                {
                    struct __attribute__((packed, aligned(4))) Q2 { i32 v[2]; } xv;
                    xv.v[0] = (i32)(o_xq[0] | (o_xq[1] << 16)); xv.v[1] = (i32)(o_xq[2] | (o_xq[3] << 16));
                    *reinterpret_cast<Q2 *>(&pxq[i0]) = xv;
                    Q4 v;
#pragma unroll
                    for (int u = 0; u < 4; u++) v.v[u] = nw[u];
                    *reinterpret_cast<Q4 *>(&NSQ.sLPC_Q14[32 + i0]) = v;
#pragma unroll
                    for (int u = 0; u < 4; u++) v.v[u] = o_shp[u];
                    *reinterpret_cast<Q4 *>(&NSQ.sLTP_shp_Q14[shp_idx0]) = v;
#pragma unroll
                    for (int u = 0; u < 4; u++) v.v[u] = o_ltp[u];
                    *reinterpret_cast<Q4 *>(&sLTP_Q15[ltp_idx0]) = v;
                }
                // the histories move by the four samples at once: [lpc_old (older 16) | lp (newer 16, newest first)], the delay line up to its order
#pragma unroll
                for (int j = 0; j < 16; j++) lpc_old[j] = j < 12 ? lpc_old[j + 4] : lp[27 - j];
#pragma unroll
                for (int j = 15; j >= 4; j--) lp[j] = lp[j - 4];
#pragma unroll
                for (int j = 0; j < 4; j++) lp[j] = nw[3 - j];
#pragma unroll
                for (int j = 15; j >= 0; j--) {
                    const i32 moved = j < 4 ? sv[3 - j] : ar[j - 4];
                    ar[j] = j < shapingLPCOrder ? moved : ar[j];
                }
            }
            continue;                                         // next subframe
        }
        const bool ahead = lag >= 8;
        i32 xn = x_Q3[0];
        // The five per-sample outputs (pulse, xq, sLPC_Q14, sLTP_shp_Q14, sLTP_Q15) are collected for four samples and
        // written with one store each (a store instruction costs the wavefront one cache line per lane whatever its width).
        // Nothing reads them back within the group when lag >= 8; otherwise the group is one sample.
        struct __attribute__((packed, aligned(4))) V16 { i32 x, y, z, w; };
        struct __attribute__((packed, aligned(4))) V8 { i32 x, y; };
        const int G = (ahead || lag <= 0) ? 4 : 1;            // no pitch lag: nothing is read back at all
        for (int i0 = 0; i0 < subfr_length; i0 += G) {
          i32 o_lpc[4] = {0, 0, 0, 0}, o_shp[4] = {0, 0, 0, 0}, o_ltp[4] = {0, 0, 0, 0};
          u32 o_xq[4] = {0, 0, 0, 0}, o_pl[4] = {0, 0, 0, 0};
          const int shp_idx0 = sLTP_shp_buf_idx, ltp_idx0 = sLTP_buf_idx;
          int cnt = 0;
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int i = i0 + u;
            if (u >= G || i >= subfr_length) break;
            cnt = u + 1;
            const i32 xcur = xn;
            xn = x_Q3[i + 1 < subfr_length ? i + 1 : i];
            if (ahead) {
                if (voiced) pn = sLTP_Q15[pred_lag + 1];
                sn = NSQ.sLTP_shp_Q14[shp_lag + 1];
            }
            rand_seed = (i32)(907633515u + (u32)rand_seed * 196314165u);
            i32 LPC_pred_Q10 = predictLPCOrder >> 1;
#pragma unroll
            for (int j = 0; j < 10; j++) LPC_pred_Q10 = s_smlawb(LPC_pred_Q10, lp[j], a12[j]);
            if (predictLPCOrder == 16) {
#pragma unroll
                for (int j = 10; j < 16; j++) LPC_pred_Q10 = s_smlawb(LPC_pred_Q10, lp[j], a12[j]);
            }
            i32 LTP_pred_Q13 = 0;
            if (voiced) {
                LTP_pred_Q13 = 2;
                LTP_pred_Q13 = s_smlawb(LTP_pred_Q13, pt[0], b0);
                LTP_pred_Q13 = s_smlawb(LTP_pred_Q13, pt[1], b1);
                LTP_pred_Q13 = s_smlawb(LTP_pred_Q13, pt[2], b2);
                LTP_pred_Q13 = s_smlawb(LTP_pred_Q13, pt[3], b3);
                LTP_pred_Q13 = s_smlawb(LTP_pred_Q13, pt[4], b4);
            }
            // noise shape feedback: the sAR2 delay line shifts by one (NSQ.c:262-279)
            i32 tmp2 = lp[0], tmp1 = ar[0];
            ar[0] = tmp2;
            i32 n_AR_Q12 = shapingLPCOrder >> 1;
            n_AR_Q12 = s_smlawb(n_AR_Q12, tmp2, ar13[0]);
#pragma unroll
            for (int j = 2; j < 16; j += 2) {
                if (j < shapingLPCOrder) {
                    tmp2 = ar[j - 1];
                    ar[j - 1] = tmp1;
                    n_AR_Q12 = s_smlawb(n_AR_Q12, tmp1, ar13[j - 1]);
                    tmp1 = ar[j];
                    ar[j] = tmp2;
                    n_AR_Q12 = s_smlawb(n_AR_Q12, tmp2, ar13[j]);
                }
            }
            {
                // ar[shapingLPCOrder-1] = tmp1 with a uniform (batch-wide) order: 10 or 16 in practice
                i32 last_coef = 0;
#pragma unroll
                for (int j = 1; j < 16; j += 2)
                    if (j == shapingLPCOrder - 1) { ar[j] = tmp1; last_coef = ar13[j]; }
                n_AR_Q12 = s_smlawb(n_AR_Q12, tmp1, last_coef);
            }
            n_AR_Q12 = shl32(n_AR_Q12, 1);
            n_AR_Q12 = s_smlawb(n_AR_Q12, sLF_AR_shp_Q14, Tilt_Q14);
            i32 n_LF_Q12 = s_smulwb(shp_prev, LF_shp_Q14);
            n_LF_Q12 = s_smlawt(n_LF_Q12, sLF_AR_shp_Q14, LF_shp_Q14);
            tmp1 = s_subw(shl32(LPC_pred_Q10, 2), n_AR_Q12);
            tmp1 = s_subw(tmp1, n_LF_Q12);
            if (lag > 0) {
                i32 n_LTP_Q13 = s_smulwb(s_addw(st[0], st[2]), HarmShapeFIRPacked_Q14);
                n_LTP_Q13 = s_smlawt(n_LTP_Q13, st[1], HarmShapeFIRPacked_Q14);
                n_LTP_Q13 = shl32(n_LTP_Q13, 1);
                tmp2 = s_subw(LTP_pred_Q13, n_LTP_Q13);
                tmp1 = s_addw(tmp2, shl32(tmp1, 1));
                tmp1 = s_rshift_round(tmp1, 3);
            } else {
                tmp1 = s_rshift_round(tmp1, 2);
            }
            i32 r_Q10 = s_subw(s_smulww(xcur, inv_gain_Q23), tmp1);
            if (rand_seed < 0) r_Q10 = (i32)(0u - (u32)r_Q10);
            r_Q10 = s_limit(r_Q10, -(31 << 10), 30 << 10);
            i32 q1_Q10 = r_Q10 - offset_Q10, q2_Q10, rd1_Q20, rd2_Q20;
            const i32 q1_Q0 = q1_Q10 >> 10;
            if (q1_Q0 > 0) {
                q1_Q10 = shl32(q1_Q0, 10) - 80 + offset_Q10;
                q2_Q10 = q1_Q10 + 1024;
                rd1_Q20 = s_smulbb(q1_Q10, Lambda_Q10);
                rd2_Q20 = s_smulbb(q2_Q10, Lambda_Q10);
            } else if (q1_Q0 == 0) {
                q1_Q10 = offset_Q10;
                q2_Q10 = q1_Q10 + (1024 - 80);
                rd1_Q20 = s_smulbb(q1_Q10, Lambda_Q10);
                rd2_Q20 = s_smulbb(q2_Q10, Lambda_Q10);
            } else if (q1_Q0 == -1) {
                q2_Q10 = offset_Q10;
                q1_Q10 = q2_Q10 - (1024 - 80);
                rd1_Q20 = s_smulbb(-q1_Q10, Lambda_Q10);
                rd2_Q20 = s_smulbb(q2_Q10, Lambda_Q10);
            } else {
                q1_Q10 = shl32(q1_Q0, 10) + 80 + offset_Q10;
                q2_Q10 = q1_Q10 + 1024;
                rd1_Q20 = s_smulbb(-q1_Q10, Lambda_Q10);
                rd2_Q20 = s_smulbb(-q2_Q10, Lambda_Q10);
            }
            i32 rr_Q10 = r_Q10 - q1_Q10;
            rd1_Q20 = s_addw(rd1_Q20, s_smulbb(rr_Q10, rr_Q10));
            rr_Q10 = r_Q10 - q2_Q10;
            rd2_Q20 = s_addw(rd2_Q20, s_smulbb(rr_Q10, rr_Q10));
            if (rd2_Q20 < rd1_Q20) q1_Q10 = q2_Q10;
            const i32 pulse = (i8)s_rshift_round(q1_Q10, 10);
            o_pl[u] = (u32)pulse & 0xffu;
            i32 exc_Q14 = shl32(q1_Q10, 4);
            if (rand_seed < 0) exc_Q14 = (i32)(0u - (u32)exc_Q14);
            const i32 LPC_exc_Q14 = s_addw(exc_Q14, shl32(LTP_pred_Q13, 1));
            const i32 xq_Q14 = s_addw(LPC_exc_Q14, shl32(LPC_pred_Q10, 4));
            {   // silk_SAT16(silk_RSHIFT_ROUND(silk_SMULWW(xq_Q14, Gain_Q10), 8)) in 64 bits
                i64 t = ((i64)xq_Q14 * Gain_Q10) >> 16;
                t = ((t >> 7) + 1) >> 1;
                o_xq[u] = (u32)(i32)(t > 32767 ? 32767 : (t < -32768 ? -32768 : t)) & 0xffffu;
            }
            // slide the LPC history (register shift) and record the sample in the state buffer
            {
                // the 32-sample history is [lpc_old (older 16) | lp (newer 16)]: the sample leaving lp enters lpc_old
                i32 leaving = lp[15];
#pragma unroll
                for (int j = 0; j < 15; j++) lpc_old[j] = lpc_old[j + 1];
                lpc_old[15] = leaving;
#pragma unroll
                for (int j = 15; j > 0; j--) lp[j] = lp[j - 1];
                lp[0] = xq_Q14;
            }
            o_lpc[u] = xq_Q14;
            sLF_AR_shp_Q14 = s_subw(xq_Q14, shl32(n_AR_Q12, 2));
            shp_prev = s_subw(sLF_AR_shp_Q14, shl32(n_LF_Q12, 2));
            o_shp[u] = shp_prev;
            o_ltp[u] = shl32(LPC_exc_Q14, 1);
            sLTP_shp_buf_idx++;
            sLTP_buf_idx++;
            rand_seed = (i32)((u32)rand_seed + (u32)pulse);
            // slide the taps
            pred_lag++;
            shp_lag++;
            pt[4] = pt[3]; pt[3] = pt[2]; pt[2] = pt[1]; pt[1] = pt[0]; pt[0] = pn;
            st[2] = st[1]; st[1] = st[0]; st[0] = sn;
          }
          if (cnt == 4) {
              *reinterpret_cast<u32 *>(&pulses[k * subfr_length + i0]) = o_pl[0] | (o_pl[1] << 8) | (o_pl[2] << 16) | (o_pl[3] << 24);
              V8 xv; xv.x = (i32)(o_xq[0] | (o_xq[1] << 16)); xv.y = (i32)(o_xq[2] | (o_xq[3] << 16));
              *reinterpret_cast<V8 *>(&pxq[i0]) = xv;
              V16 v; v.x = o_lpc[0]; v.y = o_lpc[1]; v.z = o_lpc[2]; v.w = o_lpc[3];
              *reinterpret_cast<V16 *>(&NSQ.sLPC_Q14[32 + i0]) = v;
              v.x = o_shp[0]; v.y = o_shp[1]; v.z = o_shp[2]; v.w = o_shp[3];
              *reinterpret_cast<V16 *>(&NSQ.sLTP_shp_Q14[shp_idx0]) = v;
              v.x = o_ltp[0]; v.y = o_ltp[1]; v.z = o_ltp[2]; v.w = o_ltp[3];
              *reinterpret_cast<V16 *>(&sLTP_Q15[ltp_idx0]) = v;
          } else {
#pragma unroll
              for (int u = 0; u < 4; u++) {
                  if (u >= cnt) break;
                  pulses[k * subfr_length + i0 + u] = (i8)o_pl[u];
                  pxq[i0 + u] = (i16)o_xq[u];
                  NSQ.sLPC_Q14[32 + i0 + u] = o_lpc[u];
                  NSQ.sLTP_shp_Q14[shp_idx0 + u] = o_shp[u];
                  sLTP_Q15[ltp_idx0 + u] = o_ltp[u];
              }
          }
          if (G == 1) {
              // tiny lags (never produced by the SILK pitch analysis, whose lags are >= 2 ms): a step may read what the step
              // before it stored, so the taps are re-read from memory after every (single-sample) group
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
              __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
              if (voiced) {
#pragma unroll
                  for (int j = 0; j < 5; j++) pt[j] = sLTP_Q15[pred_lag - j];
              }
              if (lag > 0) {
#pragma unroll
                  for (int j = 0; j < 3; j++) st[j] = NSQ.sLTP_shp_Q14[shp_lag - j];
              }
          }
        }
    }
    // ---- state write-back ----
#pragma unroll
    for (int j = 0; j < 16; j++) {
        NSQ.sAR2_Q14[j] = ar[j];
        NSQ.sLPC_Q14[j] = lpc_old[j];
        NSQ.sLPC_Q14[31 - j] = lp[j];
    }
    NSQ.sLF_AR_shp_Q14 = sLF_AR_shp_Q14;
    NSQ.lagPrev = in.pitchL[nb_subfr - 1];
    NSQ.sLTP_buf_idx = sLTP_buf_idx;
    NSQ.sLTP_shp_buf_idx = sLTP_shp_buf_idx;
    NSQ.rand_seed = rand_seed;
    NSQ.prev_gain_Q16 = prev_gain_Q16;
    NSQ.rewhite_flag = rewhite_flag;
    // silk_memmove of the histories (NSQ.c:176-177): ascending copy is safe (dst < src). The record is only 4-byte
    // aligned (4 380 bytes), so the copy moves 16 bytes at a time through a 4-byte-aligned vector type.
    if ((ltp_mem_length & 7) == 0 && (frame_length & 1) == 0) {
        struct __attribute__((packed, aligned(4))) V16 { i32 x, y, z, w; };
        // eight 16-byte pieces are loaded before the first of them is stored (one round trip per eight instead of per piece); a piece
        // is stored below everything a later batch loads (the source runs frame_length elements ahead of the destination)
        auto move_run = [&](V16 *d, const V16 *sc, const int n) {
            int i = 0;
            for (; i + 8 <= n; i += 8) {
                V16 v[8];
#pragma unroll
                for (int c = 0; c < 8; c++) v[c] = sc[i + c];
#pragma unroll
                for (int c = 0; c < 8; c++) d[i + c] = v[c];
            }
            for (; i < n; i++) { const V16 v = sc[i]; d[i] = v; }
        };
        move_run(reinterpret_cast<V16 *>(NSQ.xq), reinterpret_cast<const V16 *>(NSQ.xq + frame_length), ltp_mem_length / 8);
        move_run(reinterpret_cast<V16 *>(NSQ.sLTP_shp_Q14), reinterpret_cast<const V16 *>(NSQ.sLTP_shp_Q14 + frame_length), ltp_mem_length / 4);
    } else {
        for (int i = 0; i < ltp_mem_length; i++) NSQ.xq[i] = NSQ.xq[i + frame_length];
        for (int i = 0; i < ltp_mem_length; i++) NSQ.sLTP_shp_Q14[i] = NSQ.sLTP_shp_Q14[i + frame_length];
    }
}

}  // namespace ca

using namespace ca;

extern "C" int opusgpu_silk_burg_modified_batch(const opusgpu_burg_in *d_in, opusgpu_burg_out *d_out, int n, void *stream)
{
    if (n < 0) return OPUSGPU_BAD_ARG;
    if (n == 0) return OPUSGPU_OK;
    if (!d_in || !d_out) return OPUSGPU_BAD_ARG;
    int *bad = opusgpu_bad_record_counter();
    if (!bad) return OPUSGPU_ALLOC_FAIL;
    hipLaunchKernelGGL(silk_burg_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, d_in, d_out, n, bad);
    return opusgpu_check_launch();
}

extern "C" size_t opusgpu_silk_nsq_workspace_bytes(int n) { return n <= 0 ? 0 : (size_t)n * sizeof(NsqScratch); }

extern "C" int opusgpu_silk_nsq_batch(const opusgpu_nsq_in *d_in, opusgpu_nsq_state *d_state, opusgpu_nsq_out *d_out, int n,
                                      void *d_workspace, size_t workspace_bytes, void *stream)
{
    if (n < 0) return OPUSGPU_BAD_ARG;
    if (n == 0) return OPUSGPU_OK;
    if (!d_in || !d_state || !d_out || !d_workspace) return OPUSGPU_BAD_ARG;
    if (workspace_bytes < (size_t)n * sizeof(NsqScratch)) return OPUSGPU_BUFFER_TOO_SMALL;
    int *bad = opusgpu_bad_record_counter();
    if (!bad) return OPUSGPU_ALLOC_FAIL;
    hipLaunchKernelGGL(silk_nsq_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, d_in, d_state, d_out,
                       (NsqScratch *)d_workspace, n, bad, (const int *)nullptr);
    return opusgpu_check_launch();
}

// records d_rows[0 .. m) of the arrays, in place (workspace sized for the whole arrays): silk_chain.hip's bitrate loop
extern "C" int opusgpu_silk_nsq_rows(const opusgpu_nsq_in *d_in, opusgpu_nsq_state *d_state, opusgpu_nsq_out *d_out, const int *d_rows, int m,
                                     void *d_workspace, hipStream_t stream)
{
    if (m <= 0) return m < 0 ? OPUSGPU_BAD_ARG : OPUSGPU_OK;
    int *bad = opusgpu_bad_record_counter();
    if (!bad) return OPUSGPU_ALLOC_FAIL;
    hipLaunchKernelGGL(silk_nsq_kernel, dim3((m + 63) / 64), dim3(64), 0, stream, d_in, d_state, d_out, (NsqScratch *)d_workspace, m, bad, d_rows);
    return opusgpu_check_launch();
}

// Per-call hook with the reference's own signature (macro silk_burg_modified -> silk_burg_modified_c,
// opus-fix/silk/SigProc_FIX.h:601-602, fixed/burg_modified_FIX.c:45): host pointers, one record, synchronous.
// Plumbing / parity only -- a launch per call is latency-bound; throughput comes from the batch entry point.
extern "C" void opusgpu_silk_burg_modified_c(int32_t *res_nrg, int *res_nrg_Q, int32_t A_Q16[], const int16_t x[],
                                             const int32_t minInvGain_Q30, const int subfr_length, const int nb_subfr,
                                             const int D, int arch)
{
    (void)arch;
    if (!res_nrg || !res_nrg_Q || !A_Q16 || !x || D < 1 || D > OPUSGPU_SILK_MAX_ORDER || nb_subfr < 1 || subfr_length <= D ||
        subfr_length * nb_subfr > OPUSGPU_SILK_BURG_MAX_X) {
        opusgpu_set_last_error(OPUSGPU_BAD_ARG);
        return;
    }
    opusgpu_burg_in h_in;
    memset(&h_in, 0, sizeof(h_in));
    memcpy(h_in.x, x, sizeof(int16_t) * (size_t)subfr_length * nb_subfr);
    h_in.minInvGain_Q30 = minInvGain_Q30; h_in.subfr_length = subfr_length; h_in.nb_subfr = nb_subfr; h_in.D = D;
    opusgpu_burg_in *d_in = nullptr;
    opusgpu_burg_out *d_out = nullptr, h_out;
    if (hipMalloc(&d_in, sizeof(h_in)) != hipSuccess || hipMalloc(&d_out, sizeof(h_out)) != hipSuccess) {
        opusgpu_set_last_error(OPUSGPU_ALLOC_FAIL);
        if (d_in) (void)hipFree(d_in);
        return;
    }
    int rc = opusgpu_copy(d_in, &h_in, sizeof(h_in), hipMemcpyHostToDevice);
    if (rc == OPUSGPU_OK) rc = opusgpu_silk_burg_modified_batch(d_in, d_out, 1, nullptr);
    if (rc == OPUSGPU_OK && hipMemcpy(&h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost) != hipSuccess) rc = OPUSGPU_INTERNAL_ERROR;
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    opusgpu_set_last_error(rc);
    if (rc != OPUSGPU_OK) return;
    *res_nrg = h_out.res_nrg;
    *res_nrg_Q = h_out.res_nrg_Q;
    memcpy(A_Q16, h_out.A_Q16, sizeof(int32_t) * (size_t)D);
}

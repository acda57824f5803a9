// silk_nsq_del_dec.hip -- silk_NSQ_del_dec(), the noise-shaping quantizer of the reference at complexity >= 4,
// batched over function-boundary records (include/opusgpu_silk.h: opusgpu_nsq_dd_in / opusgpu_nsq_state / opusgpu_nsq_dd_out).
//
//   silk_NSQ_del_dec_c                    opus-fix/silk/NSQ_del_dec.c:112-318
//   silk_noise_shape_quantizer_del_dec    opus-fix/silk/NSQ_del_dec.c:324-630
//   silk_nsq_del_dec_scale_states         opus-fix/silk/NSQ_del_dec.c:632-724
//
// Mapping: the quantizer keeps up to four candidate paths ("delayed-decision states") per record, all advanced by
// the same per-sample filter arithmetic and then compared with each other. FOUR ADJACENT LANES OWN ONE RECORD, lane k
// of the quad = path k: the per-path work (16-tap prediction, warped 16-tap shaping, two quantization candidates) is
// the parallel part, the per-sample tournament between the paths (winner, expiry, worst-of-first / best-of-second
// replacement) is a handful of quad broadcasts (DPP quad_perm), and a wavefront advances 16 records.
//   * per path, in registers: the 16 newest sLPC_Q14 samples (a shift register), sAR2_Q14[16], LF_AR, Seed, SeedInit, RD;
//   * per path, in LDS ([row][lane], lanes rotated per row): the five 32-deep decision rings RandState / Q / Xq /
//     Pred / Shape (40 KB per wavefront), with one level of indirection: a path's entry for ring slot t lies in the column of
//     the path that WROTE it, and every path carries a 64-bit map (2 bits per slot) saying which column that is;
//   * per record, in HBM: the NSQ state (xq, sLTP_shp_Q14: read at the pitch lag by all four lanes, written by the
//     winner's lane) and the scaled re-whitened prediction buffer sLTP_Q15 (workspace; its 16-bit sLTP half is unused).
// The reference's survivor copy (memcpy of the struct tail, :583-584) becomes: registers through ds_bpermute -- the map among
// them: copying it IS the copy of the 160 ring rows (a replaced path goes on reading its new parent's history where the
// parent wrote it; all four paths write slot t in the same step, each into its own column, so no entry is overwritten while a
// map still points at it).
// sLPC_Q14[0..32) of the state output is rebuilt from the winner's Xq ring: after the last subframe (length >= 32)
// both hold the last 32 xq_Q14 values (:307).
#include "silk_math.h"
#include "opusgpu_internal.h"
#include "../../include/opusgpu_silk.h"
#include "silk_validate.h"

namespace ca {

namespace dd {
enum { DELAY = 32, MASK = 31, R_RND = 0, R_Q = 32, R_XQ = 64, R_PRED = 96, R_SHAPE = 128, ROWS = 160 };
// ring element (row, lane): 64 words per row, the lane rotated by the row number -- a row read by all lanes and a column
// walked by the four lanes of a quad (the survivor copy) both spread over the banks. 160 rows x 256 B = 40 KB per
// wavefront, four workgroups per CU.
CA_DEV int ridx(int row, int lanecol) { return row * 64 + ((lanecol + row) & 63); }
struct Scratch { i32 sLTP_Q15[640]; i16 sLTP[640]; };
struct Cand { i32 q, rd, xq, lf_ar, shp, exc; };

// value of quad lane J (compile-time) / j (run-time, quad-uniform) in every lane of the quad
template <int J> CA_DEV i32 qb(i32 v) { return __builtin_amdgcn_update_dpp(0, v, J * 0x55, 0xf, 0xf, false); }
CA_DEV i32 qsel(i32 v, int quad_base, int j) { return __shfl(v, quad_base + j, 64); }
// Orders the quad's traffic through LDS and through global memory. The four lanes of a record sit in ONE wavefront
// (a workgroup is one wavefront here), whose LDS and vector-memory operations are performed in program order, so
// wavefront scope is enough: the fence constrains the compiler and costs no s_waitcnt.
CA_DEV void quad_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// column (0 .. 3) of ring slot t in a path's map; the map's entry for slot t set to k
CA_DEV int mget(u32 m0, u32 m1, int t) { return (int)(((t & 16) ? m1 : m0) >> ((t & 15) << 1)) & 3; }
CA_DEV void mset(u32 &m0, u32 &m1, int t, int k)
{
    const u32 sh = (u32)(t & 15) << 1, keep = ~(3u << sh), v = (u32)k << sh;
    const u32 a = (m0 & keep) | v, b = (m1 & keep) | v;
    if (t & 16) m1 = b; else m0 = a;
}
CA_DEV i16 sat16(i32 v) { return (i16)(v > 32767 ? 32767 : (v < -32768 ? -32768 : v)); }
}  // namespace dd

__global__ __launch_bounds__(64) void silk_nsq_del_dec_kernel(const opusgpu_nsq_dd_in *__restrict__ recs, opusgpu_nsq_state *states,
                                                              opusgpu_nsq_dd_out *__restrict__ outs, dd::Scratch *ws, int n_rec,
                                                              int *__restrict__ bad_records, const int *__restrict__ rows)
{
    using namespace dd;
    __shared__ i32 ring[ROWS * 64];
    const int ln = threadIdx.x, k = ln & 3, quad = ln & ~3;
    int r = blockIdx.x * 16 + (ln >> 2);
    if (r >= n_rec) return;                       // whole quads leave together
    if (rows) r = rows[r];                        // the bitrate loop's second passes: a list of frames, in place
    const opusgpu_nsq_in &in = recs[r].base;
    const int nst = recs[r].nStatesDelayedDecision, warping_Q16 = recs[r].warping_Q16;
    opusgpu_nsq_state &NSQ = states[r];
    if (!nsq_dd_record_ok(recs[r], NSQ.lagPrev)) {            // the same verdict in all four lanes of the quad: they leave together
        if (k == 0) {
            for (int e = 0; e < OPUSGPU_SILK_MAX_FRAME; e++) outs[r].pulses[e] = 0;
            outs[r].Seed = in.Seed;
            atomicAdd(bad_records, 1);
        }
        return;
    }
    i32 *sLTP_Q15 = ws[r].sLTP_Q15;
    const int nb_subfr = in.nb_subfr, L = in.subfr_length, frame_length = in.frame_length;
    const int ltp_mem = in.ltp_mem_length, pord = in.predictLPCOrder, sord = in.shapingLPCOrder;
    const int voiced = in.signalType == 2;
    const int Lambda_Q10 = in.Lambda_Q10;
    i8 *pulses = outs[r].pulses;

    for (int e = 0; e < ROWS; e++) ring[ridx(e, ln)] = 0;
    i32 lp[16], ar[16];
#pragma unroll
    for (int j = 0; j < 16; j++) { lp[j] = NSQ.sLPC_Q14[31 - j]; ar[j] = NSQ.sAR2_Q14[j]; }
    i32 seed = (k + in.Seed) & 3, seed0 = seed, rd = 0, lf_ar = NSQ.sLF_AR_shp_Q14;
    u32 m0 = 0x55555555u * (u32)k, m1 = m0;        // every slot of the (zeroed) rings in the path's own column
    ring[ridx(R_SHAPE + 0, ln)] = NSQ.sLTP_shp_Q14[ltp_mem - 1];
    i32 prev_gain_Q16 = NSQ.prev_gain_Q16;
    int lag = NSQ.lagPrev;
    const int offset_Q10 = voiced ? (in.quantOffsetType ? 100 : 32) : (in.quantOffsetType ? 240 : 100);   // silk/tables_other.c:95-97
    int smpl = 0;
    int delay = L < DELAY ? L : DELAY;
    if (voiced) {
        for (int s = 0; s < nb_subfr; s++) delay = imin(delay, in.pitchL[s] - 5 / 2 - 1);
    } else if (lag > 0) {
        delay = imin(delay, lag - 5 / 2 - 1);
    }
    const int interp = in.NLSFInterpCoef_Q2 == 4 ? 0 : 1;
    int shp_idx = ltp_mem, ltp_idx = ltp_mem;     // NSQ->sLTP_shp_buf_idx, NSQ->sLTP_buf_idx
    int rewhite = 0, subfr = 0;
    quad_fence();

    for (int sf = 0; sf < nb_subfr; sf++) {
        const i16 *A_Q12 = &in.PredCoef_Q12[((sf >> 1) | (1 - interp)) * 16];
        const i16 *B_Q14 = &in.LTPCoef_Q14[sf * 5];
        const i16 *AR_Q13 = &in.AR2_Q13[sf * 16];
        const i32 *x_Q3 = &in.x_Q3[sf * L];
        i8 *pl = pulses + sf * L;
        i16 *pxq = &NSQ.xq[ltp_mem + sf * L];
        i32 harm = in.HarmShapeGain_Q14[sf] >> 2;
        harm |= shl32(in.HarmShapeGain_Q14[sf] >> 1, 16);
        rewhite = 0;
        if (voiced) {
            lag = in.pitchL[sf];
            if ((sf & (3 - (interp << 1))) == 0) {
                if (sf == 2) {
                    // NSQ_del_dec.c:190-221: before the prediction filter changes, flush the surviving path
                    int w = 0;
                    i32 best = qb<0>(rd);
                    { i32 v = qb<1>(rd); if (nst > 1 && v < best) { best = v; w = 1; } }
                    { i32 v = qb<2>(rd); if (nst > 2 && v < best) { best = v; w = 2; } }
                    { i32 v = qb<3>(rd); if (nst > 3 && v < best) { best = v; w = 3; } }
                    if (k != w) rd = s_addw(rd, 0x7FFFFFFF >> 4);
                    const u32 wm0 = (u32)qsel((i32)m0, quad, w), wm1 = (u32)qsel((i32)m1, quad, w);
                    for (int i = k; i < delay; i += 4) {
                        const int last = (smpl + delay - 1 - i) & MASK, wc = quad + mget(wm0, wm1, last);
                        pl[i - delay] = (i8)s_rshift_round(ring[ridx(R_Q + last, wc)], 10);
                        pxq[i - delay] = sat16(s_rshift_round(s_smulww(ring[ridx(R_XQ + last, wc)], in.Gains_Q16[1]), 14));
                        NSQ.sLTP_shp_Q14[shp_idx - delay + i] = ring[ridx(R_SHAPE + last, wc)];
                    }
                    subfr = 0;
                    quad_fence();
                }
                // re-whitening: silk_LPC_analysis_filter (celt_fir form), the output samples shared out over the quad
                const int start_idx = ltp_mem - lag - pord - 5 / 2;
                const i16 *inp = &NSQ.xq[start_idx + sf * L];
                const int len = ltp_mem - start_idx;
                {
                    // every lane of the quad takes a contiguous quarter of the outputs; the pord previous input samples travel in a
                    // register window, so each sample of xq is read once per lane that needs it. The outputs leave scaled, as
                    // silk_nsq_del_dec_scale_states scales exactly these lag + 2 values (:651-660: sLTP_Q15[i] = SMULWB(inv_gain_Q31,
                    // sLTP[i])): the 16-bit buffer in between is never touched; four inputs per load, four results per store
                    const int chunk = (len - pord + 3) >> 2, i0 = pord + k * chunk, i1 = imin(len, i0 + chunk);
                    i32 ig_Q31 = s_inverse32_varq(in.Gains_Q16[sf] > 1 ? in.Gains_Q16[sf] : 1, 47);
                    if (sf == 0) ig_Q31 = shl32(s_smulwb(ig_Q31, in.LTP_scale_Q14), 2);
                    i32 *outq = &sLTP_Q15[start_idx];
                    i32 nA[16], w[16];
#pragma unroll
                    for (int m = 0; m < 16; m++) {
                        nA[m] = m < pord ? (i32)(i16)(-A_Q12[m]) : 0;
                        w[m] = (m < pord && i0 < i1) ? (i32)inp[i0 - 1 - m] : 0;
                    }
                    struct __attribute__((packed, aligned(2))) H4 { i16 v[4]; };
                    struct __attribute__((packed, aligned(4))) W4 { i32 v[4]; };
                    int ix = i0;
                    for (; ix + 4 <= i1; ix += 4) {
                        const H4 xi4 = *reinterpret_cast<const H4 *>(&inp[ix]);
                        W4 o;
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            i32 sum = 0;
#pragma unroll
                            for (int m = 0; m < 16; m++) sum = s_addw(sum, __mul24(nA[m], w[m]));
                            const i32 xi = (i32)xi4.v[u];
                            o.v[u] = s_smulwb(ig_Q31, (i32)sat16(xi + pshr32(sum, 12)));
#pragma unroll
                            for (int m = 15; m > 0; m--) w[m] = w[m - 1];
                            w[0] = xi;
                        }
                        *reinterpret_cast<W4 *>(&outq[ix]) = o;
                    }
                    for (; ix < i1; ix++) {
                        i32 sum = 0;
#pragma unroll
                        for (int m = 0; m < 16; m++) sum = s_addw(sum, __mul24(nA[m], w[m]));
                        const i32 xi = (i32)inp[ix];
                        outq[ix] = s_smulwb(ig_Q31, (i32)sat16(xi + pshr32(sum, 12)));
#pragma unroll
                        for (int m = 15; m > 0; m--) w[m] = w[m - 1];
                        w[0] = xi;
                    }
                }
                ltp_idx = ltp_mem;
                rewhite = 1;
                quad_fence();
            }
        }
        // ---- silk_nsq_del_dec_scale_states (NSQ_del_dec.c:632-724) ----
        const i32 gain = in.Gains_Q16[sf];
        const i32 inv_gain_Q31 = s_inverse32_varq(gain > 1 ? gain : 1, 47);
        const i32 adj = gain != prev_gain_Q16 ? s_div32_varq(prev_gain_Q16, gain, 16) : (i32)1 << 16;
        const i32 inv_gain_Q23 = s_rshift_round(inv_gain_Q31, 8);
        prev_gain_Q16 = gain;
        {
            const int lg = in.pitchL[sf];
            // (rewhite: sLTP_Q15[idx - lag - 2 .. idx) was scaled where it was produced, above)
            if (adj != (i32)1 << 16) {
                {
                    // every lane of the quad takes four consecutive values per access (16 bytes; the record is 4-byte aligned)
                    struct __attribute__((packed, aligned(4))) Q4 { i32 v[4]; };
                    const int i0 = shp_idx - ltp_mem, body = ltp_mem & ~15;
                    int i = i0 + 4 * k;
                    for (; i + 48 < i0 + body; i += 64) {               // four accesses in flight before the first store
                        Q4 q[4];
#pragma unroll
                        for (int c = 0; c < 4; c++) q[c] = *reinterpret_cast<const Q4 *>(&NSQ.sLTP_shp_Q14[i + 16 * c]);
#pragma unroll
                        for (int c = 0; c < 4; c++) {
#pragma unroll
                            for (int u = 0; u < 4; u++) q[c].v[u] = s_smulww(adj, q[c].v[u]);
                            *reinterpret_cast<Q4 *>(&NSQ.sLTP_shp_Q14[i + 16 * c]) = q[c];
                        }
                    }
                    for (; i < i0 + body; i += 16) {
                        Q4 q = *reinterpret_cast<const Q4 *>(&NSQ.sLTP_shp_Q14[i]);
#pragma unroll
                        for (int u = 0; u < 4; u++) q.v[u] = s_smulww(adj, q.v[u]);
                        *reinterpret_cast<Q4 *>(&NSQ.sLTP_shp_Q14[i]) = q;
                    }
                    for (int i = i0 + body + k; i < shp_idx; i += 4) NSQ.sLTP_shp_Q14[i] = s_smulww(adj, NSQ.sLTP_shp_Q14[i]);
                }
                if (voiced && rewhite == 0) {
                    int i = ltp_idx - lg - 5 / 2 + k;
                    const int end = ltp_idx - delay;
                    for (; i + 12 < end; i += 16) {
                        i32 q[4];
#pragma unroll
                        for (int c = 0; c < 4; c++) q[c] = sLTP_Q15[i + 4 * c];
#pragma unroll
                        for (int c = 0; c < 4; c++) sLTP_Q15[i + 4 * c] = s_smulww(adj, q[c]);
                    }
                    for (; i < end; i += 4) sLTP_Q15[i] = s_smulww(adj, sLTP_Q15[i]);
                }
                lf_ar = s_smulww(adj, lf_ar);
#pragma unroll
                for (int j = 0; j < 16; j++) { lp[j] = s_smulww(adj, lp[j]); ar[j] = s_smulww(adj, ar[j]); }
#pragma unroll 8
                for (int i = 0; i < DELAY; i++) {
                    ring[ridx(R_PRED + i, ln)] = s_smulww(adj, ring[ridx(R_PRED + i, ln)]);
                    ring[ridx(R_SHAPE + i, ln)] = s_smulww(adj, ring[ridx(R_SHAPE + i, ln)]);
                }
            }
            quad_fence();
        }
        // ---- silk_noise_shape_quantizer_del_dec (NSQ_del_dec.c:324-630) ----
        // delayedGain_Q10[] of the reference (:596, :610): the sample released at step i is `delay` (<= subframe length) old,
        // so its gain is this subframe's or the previous one's -- no ring needed
        const i32 Gain_Q10 = gain >> 6, prev_Gain_Q10 = sf > 0 ? in.Gains_Q16[sf - 1] >> 6 : 0;
        const int Tilt_Q14 = in.Tilt_Q14[sf];
        const i32 LF_shp_Q14 = in.LF_shp_Q14[sf];
        const i32 *shp_lag = &NSQ.sLTP_shp_Q14[shp_idx - lag + 3 / 2];
        const i32 *pred_lag = &sLTP_Q15[ltp_idx - lag + 5 / 2];
        // the filter coefficients of the subframe in registers (zero beyond the order: a zero tap adds nothing)
        i32 cA[16], cAR[16], cB[5];
#pragma unroll
        for (int j = 0; j < 16; j++) { cA[j] = j < pord ? (i32)A_Q12[j] : 0; cAR[j] = j < sord ? (i32)AR_Q13[j] : 0; }
#pragma unroll
        for (int j = 0; j < 5; j++) cB[j] = B_Q14[j];
        const i32 cAR_last = AR_Q13[sord - 1];
        // The pitch-lag taps slide by one sample per step: pt[j] = pred_lag[-j], st[j] = shp_lag[-j] are kept in registers
        // and only the newest tap is loaded. What a step stores (position idx - delay) is read as a newest tap no earlier
        // than one step later (prediction: idx - lag + 2, delay <= lag - 3) or two steps later (shaping: idx - lag + 1), so
        // the next shaping tap is always fetched a whole step ahead, the next prediction tap when delay <= lag - 4.
        i32 pt[5] = {0, 0, 0, 0, 0}, st[3] = {0, 0, 0}, pn = 0, sn = 0;
        if (voiced) {
#pragma unroll
            for (int j = 0; j < 5; j++) pt[j] = pred_lag[-j];
        }
        if (lag > 0) {
#pragma unroll
            for (int j = 0; j < 3; j++) st[j] = shp_lag[-j];
        }
        const bool pred_ahead = delay <= lag - 4;
        i32 xn = x_Q3[0];
        for (int i = 0; i < L; i++) {
            const i32 xcur = xn;
            xn = x_Q3[i + 1 < L ? i + 1 : i];
            if (voiced && pred_ahead) pn = pred_lag[1];
            if (lag > 0) sn = shp_lag[1];
            // common to the paths of a record (computed by each of its lanes)
            i32 LTP_pred_Q14 = 0, n_LTP_Q14 = 0;
            if (voiced) {
                LTP_pred_Q14 = 2;
#pragma unroll
                for (int j = 0; j < 5; j++) LTP_pred_Q14 = s_smlawb(LTP_pred_Q14, pt[j], cB[j]);
                LTP_pred_Q14 = shl32(LTP_pred_Q14, 1);
            }
            if (lag > 0) {
                n_LTP_Q14 = s_smulwb(s_addw(st[0], st[2]), harm);
                n_LTP_Q14 = s_smlawt(n_LTP_Q14, st[1], harm);
                n_LTP_Q14 = s_subw(LTP_pred_Q14, shl32(n_LTP_Q14, 2));
            }
            const i32 x_sc_Q10 = s_smulww(xcur, inv_gain_Q23);
            // this lane's path
            seed = (i32)(907633515u + (u32)seed * 196314165u);                  // silk_RAND
            i32 LPC_pred_Q14 = pord >> 1;
#pragma unroll
            for (int j = 0; j < 16; j++) LPC_pred_Q14 = s_smlawb(LPC_pred_Q14, lp[j], cA[j]);
            LPC_pred_Q14 = shl32(LPC_pred_Q14, 4);
            i32 tmp2 = s_smlawb(lp[0], ar[0], warping_Q16);
            i32 tmp1 = s_smlawb(ar[0], s_subw(ar[1], tmp2), warping_Q16);
            ar[0] = tmp2;
            i32 n_AR_Q14 = sord >> 1;
            n_AR_Q14 = s_smlawb(n_AR_Q14, tmp2, cAR[0]);
#pragma unroll
            for (int j = 2; j < 16; j += 2) {
                if (j < sord) {
                    tmp2 = s_smlawb(ar[j - 1], s_subw(ar[j], tmp1), warping_Q16);
                    ar[j - 1] = tmp1;
                    n_AR_Q14 = s_smlawb(n_AR_Q14, tmp1, cAR[j - 1]);
                    tmp1 = s_smlawb(ar[j], s_subw(ar[j + 1], tmp2), warping_Q16);
                    ar[j] = tmp2;
                    n_AR_Q14 = s_smlawb(n_AR_Q14, tmp2, cAR[j]);
                }
            }
#pragma unroll
            for (int j = 1; j < 16; j += 2) if (j == sord - 1) ar[j] = tmp1;
            n_AR_Q14 = s_smlawb(n_AR_Q14, tmp1, cAR_last);
            n_AR_Q14 = shl32(n_AR_Q14, 1);
            n_AR_Q14 = s_smlawb(n_AR_Q14, lf_ar, Tilt_Q14);
            n_AR_Q14 = shl32(n_AR_Q14, 2);
            i32 n_LF_Q14 = s_smulwb(ring[ridx(R_SHAPE + smpl, ln)], LF_shp_Q14);
            n_LF_Q14 = s_smlawt(n_LF_Q14, lf_ar, LF_shp_Q14);
            n_LF_Q14 = shl32(n_LF_Q14, 2);
            tmp1 = s_addw(n_AR_Q14, n_LF_Q14);
            tmp2 = s_addw(n_LTP_Q14, LPC_pred_Q14);
            tmp1 = s_rshift_round(s_subw(tmp2, tmp1), 4);
            i32 r_Q10 = s_subw(x_sc_Q10, tmp1);
            if (seed < 0) r_Q10 = (i32)(0u - (u32)r_Q10);
            r_Q10 = s_limit(r_Q10, -(31 << 10), 30 << 10);
            i32 q1_Q10 = r_Q10 - offset_Q10, q2_Q10, rd1, rd2;
            const i32 q1_Q0 = q1_Q10 >> 10;
            if (q1_Q0 > 0) {
                q1_Q10 = shl32(q1_Q0, 10) - 80 + offset_Q10;
                q2_Q10 = q1_Q10 + 1024;
                rd1 = s_smulbb(q1_Q10, Lambda_Q10);
                rd2 = s_smulbb(q2_Q10, Lambda_Q10);
            } else if (q1_Q0 == 0) {
                q1_Q10 = offset_Q10;
                q2_Q10 = q1_Q10 + (1024 - 80);
                rd1 = s_smulbb(q1_Q10, Lambda_Q10);
                rd2 = s_smulbb(q2_Q10, Lambda_Q10);
            } else if (q1_Q0 == -1) {
                q2_Q10 = offset_Q10;
                q1_Q10 = q2_Q10 - (1024 - 80);
                rd1 = s_smulbb(-q1_Q10, Lambda_Q10);
                rd2 = s_smulbb(q2_Q10, Lambda_Q10);
            } else {
                q1_Q10 = shl32(q1_Q0, 10) + 80 + offset_Q10;
                q2_Q10 = q1_Q10 + 1024;
                rd1 = s_smulbb(-q1_Q10, Lambda_Q10);
                rd2 = s_smulbb(-q2_Q10, Lambda_Q10);
            }
            i32 rr = r_Q10 - q1_Q10;
            rd1 = s_addw(rd1, s_smulbb(rr, rr)) >> 10;
            rr = r_Q10 - q2_Q10;
            rd2 = s_addw(rd2, s_smulbb(rr, rr)) >> 10;
            Cand c0, c1;
            const bool q1_first = rd1 < rd2;
            c0.rd = s_addw(rd, q1_first ? rd1 : rd2);
            c1.rd = s_addw(rd, q1_first ? rd2 : rd1);
            c0.q = q1_first ? q1_Q10 : q2_Q10;
            c1.q = q1_first ? q2_Q10 : q1_Q10;
            {
                i32 e0 = shl32(c0.q, 4), e1 = shl32(c1.q, 4);
                if (seed < 0) { e0 = -e0; e1 = -e1; }
                c0.exc = s_addw(e0, LTP_pred_Q14);
                c1.exc = s_addw(e1, LTP_pred_Q14);
                c0.xq = s_addw(c0.exc, LPC_pred_Q14);
                c1.xq = s_addw(c1.exc, LPC_pred_Q14);
                c0.lf_ar = s_subw(c0.xq, n_AR_Q14);
                c1.lf_ar = s_subw(c1.xq, n_AR_Q14);
                c0.shp = s_subw(c0.lf_ar, n_LF_Q14);
                c1.shp = s_subw(c1.lf_ar, n_LF_Q14);
            }
            // ---- tournament between the paths of the record (:530-586) ----
            smpl = (smpl - 1) & MASK;
            const int last = (smpl + delay) & MASK;
            int w = 0;
            {
                i32 best = qb<0>(c0.rd);
                { i32 v = qb<1>(c0.rd); if (nst > 1 && v < best) { best = v; w = 1; } }
                { i32 v = qb<2>(c0.rd); if (nst > 2 && v < best) { best = v; w = 2; } }
                { i32 v = qb<3>(c0.rd); if (nst > 3 && v < best) { best = v; w = 3; } }
            }
            const i32 my_rand = ring[ridx(R_RND + last, quad + mget(m0, m1, last))];
            const i32 wrand = qsel(my_rand, quad, w);
            if (my_rand != wrand) {             // paths that disagree with the winner on the sample that expires now
                c0.rd = s_addw(c0.rd, 0x7FFFFFFF >> 4);
                c1.rd = s_addw(c1.rd, 0x7FFFFFFF >> 4);
            }
            int worst = 0, best2 = 0;
            {
                i32 mx = qb<0>(c0.rd), mn = qb<0>(c1.rd);
                { i32 a = qb<1>(c0.rd), b = qb<1>(c1.rd); if (nst > 1) { if (a > mx) { mx = a; worst = 1; } if (b < mn) { mn = b; best2 = 1; } } }
                { i32 a = qb<2>(c0.rd), b = qb<2>(c1.rd); if (nst > 2) { if (a > mx) { mx = a; worst = 2; } if (b < mn) { mn = b; best2 = 2; } } }
                { i32 a = qb<3>(c0.rd), b = qb<3>(c1.rd); if (nst > 3) { if (a > mx) { mx = a; worst = 3; } if (b < mn) { mn = b; best2 = 3; } } }
                const bool replace = mn < mx;
                // the survivor copy: path `worst` continues as a copy of path `best2` with its second candidate
                const int src = quad + best2;
                const bool me = replace && k == worst;
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    const i32 a = __shfl(lp[j], src, 64), b = __shfl(ar[j], src, 64);
                    if (me) { lp[j] = a; ar[j] = b; }
                }
                {
                    const i32 a = __shfl(seed, src, 64), b = __shfl(seed0, src, 64);
                    const i32 q = __shfl(c1.q, src, 64), d = __shfl(c1.rd, src, 64), x = __shfl(c1.xq, src, 64);
                    const i32 f = __shfl(c1.lf_ar, src, 64), s = __shfl(c1.shp, src, 64), e = __shfl(c1.exc, src, 64);
                    const u32 ma = (u32)__shfl((i32)m0, src, 64), mb = (u32)__shfl((i32)m1, src, 64);     // the five rings
                    if (me) { seed = a; seed0 = b; c0.q = q; c0.rd = d; c0.xq = x; c0.lf_ar = f; c0.shp = s; c0.exc = e; m0 = ma; m1 = mb; }
                }
            }
            // the winner's lane releases the sample that is `delay` old
            if (k == w && (subfr > 0 || i >= delay)) {
                const int wc = quad + mget(m0, m1, last);
                pl[i - delay] = (i8)s_rshift_round(ring[ridx(R_Q + last, wc)], 10);
                pxq[i - delay] = sat16(s_rshift_round(s_smulww(ring[ridx(R_XQ + last, wc)], i >= delay ? Gain_Q10 : prev_Gain_Q10), 8));
                NSQ.sLTP_shp_Q14[shp_idx - delay] = ring[ridx(R_SHAPE + last, wc)];
                sLTP_Q15[ltp_idx - delay] = ring[ridx(R_PRED + last, wc)];
            }
            shp_idx++;
            ltp_idx++;
            // every path advances with its first candidate
            lf_ar = c0.lf_ar;
#pragma unroll
            for (int j = 15; j > 0; j--) lp[j] = lp[j - 1];
            lp[0] = c0.xq;
            ring[ridx(R_XQ + smpl, ln)] = c0.xq;
            ring[ridx(R_Q + smpl, ln)] = c0.q;
            ring[ridx(R_PRED + smpl, ln)] = shl32(c0.exc, 1);
            ring[ridx(R_SHAPE + smpl, ln)] = c0.shp;
            seed = s_addw(seed, s_rshift_round(c0.q, 10));
            ring[ridx(R_RND + smpl, ln)] = seed;
            mset(m0, m1, smpl, k);
            rd = c0.rd;
            quad_fence();
            if (voiced && !pred_ahead) pn = pred_lag[1];
            pred_lag++;
            shp_lag++;
#pragma unroll
            for (int j = 4; j > 0; j--) pt[j] = pt[j - 1];
            pt[0] = pn;
            st[2] = st[1]; st[1] = st[0]; st[0] = sn;
        }
        subfr++;
    }
    // ---- NSQ_del_dec.c:285-317: flush the last `delay` samples of the winning path, write the state back ----
    {
        int w = 0;
        i32 best = qb<0>(rd);
        { i32 v = qb<1>(rd); if (nst > 1 && v < best) { best = v; w = 1; } }
        { i32 v = qb<2>(rd); if (nst > 2 && v < best) { best = v; w = 2; } }
        { i32 v = qb<3>(rd); if (nst > 3 && v < best) { best = v; w = 3; } }
        const i32 Gain_Q10 = in.Gains_Q16[nb_subfr - 1] >> 6;
        i8 *pl = pulses + nb_subfr * L;
        i16 *pxq = &NSQ.xq[ltp_mem + nb_subfr * L];
        const u32 wm0 = (u32)qsel((i32)m0, quad, w), wm1 = (u32)qsel((i32)m1, quad, w);
        for (int i = k; i < delay; i += 4) {
            const int last = (smpl + delay - 1 - i) & MASK, wc = quad + mget(wm0, wm1, last);
            pl[i - delay] = (i8)s_rshift_round(ring[ridx(R_Q + last, wc)], 10);
            pxq[i - delay] = sat16(s_rshift_round(s_smulww(ring[ridx(R_XQ + last, wc)], Gain_Q10), 8));
            NSQ.sLTP_shp_Q14[shp_idx - delay + i] = ring[ridx(R_SHAPE + last, wc)];
        }
        for (int m = k; m < 32; m += 4) {
            const int slot = (smpl + m) & MASK;
            NSQ.sLPC_Q14[31 - m] = ring[ridx(R_XQ + slot, quad + mget(wm0, wm1, slot))];
        }
        if (k == w) {
            outs[r].Seed = seed0;
#pragma unroll
            for (int j = 0; j < 16; j++) NSQ.sAR2_Q14[j] = ar[j];
            NSQ.sLF_AR_shp_Q14 = lf_ar;
            NSQ.lagPrev = in.pitchL[nb_subfr - 1];
            NSQ.sLTP_buf_idx = ltp_idx;
            NSQ.sLTP_shp_buf_idx = shp_idx;
            NSQ.prev_gain_Q16 = prev_gain_Q16;
            NSQ.rewhite_flag = rewhite;
        }
        quad_fence();
        // silk_memmove of the two histories by frame_length (:315-316), shared out over the quad in 16-byte pieces (the
        // record is only 4-byte aligned, hence the 4-byte-aligned vector type); a forward copy only overwrites what
        // earlier iterations have already read (the source runs frame_length >= 32 elements ahead)
        if ((ltp_mem & 7) == 0 && (frame_length & 7) == 0) {
            struct __attribute__((packed, aligned(4))) V16 { i32 x, y, z, w; };
            V16 *dq = reinterpret_cast<V16 *>(NSQ.xq);
            const V16 *sq = reinterpret_cast<const V16 *>(NSQ.xq + frame_length);
            // four pieces per lane are loaded before the first is stored: a stored piece lies below everything a later batch loads
            auto move_run = [&](V16 *d, const V16 *sc, const int n) {
                int c = k;
                for (; c + 12 < n; c += 16) {
                    V16 v[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) v[j] = sc[c + 4 * j];
#pragma unroll
                    for (int j = 0; j < 4; j++) d[c + 4 * j] = v[j];
                }
                for (; c < n; c += 4) { const V16 v = sc[c]; d[c] = v; }
            };
            move_run(dq, sq, ltp_mem / 8);
            move_run(reinterpret_cast<V16 *>(NSQ.sLTP_shp_Q14), reinterpret_cast<const V16 *>(NSQ.sLTP_shp_Q14 + frame_length), ltp_mem / 4);
        } else {
            for (int m = k; m < ltp_mem; m += 4) {
                const i16 a = NSQ.xq[m + frame_length];
                const i32 b = NSQ.sLTP_shp_Q14[m + frame_length];
                NSQ.xq[m] = a;
                NSQ.sLTP_shp_Q14[m] = b;
            }
        }
    }
}

}  // namespace ca

using namespace ca;

extern "C" size_t opusgpu_silk_nsq_del_dec_workspace_bytes(int n) { return n <= 0 ? 0 : (size_t)n * sizeof(dd::Scratch); }

extern "C" int opusgpu_silk_nsq_del_dec_batch(const opusgpu_nsq_dd_in *d_in, opusgpu_nsq_state *d_state, opusgpu_nsq_dd_out *d_out, int n,
                                              void *d_workspace, size_t workspace_bytes, void *stream)
{
    if (n < 0) return OPUSGPU_BAD_ARG;
    if (n == 0) return OPUSGPU_OK;
    if (!d_in || !d_state || !d_out || !d_workspace) return OPUSGPU_BAD_ARG;
    if (workspace_bytes < (size_t)n * sizeof(dd::Scratch)) return OPUSGPU_BUFFER_TOO_SMALL;
    int *bad = opusgpu_bad_record_counter();
    if (!bad) return OPUSGPU_ALLOC_FAIL;
    hipLaunchKernelGGL(silk_nsq_del_dec_kernel, dim3((n + 15) / 16), dim3(64), 0, (hipStream_t)stream, d_in, d_state, d_out,
                       (dd::Scratch *)d_workspace, n, bad, (const int *)nullptr);
    return opusgpu_check_launch();
}

// records d_rows[0 .. m) of the arrays, in place (workspace sized for the whole arrays): silk_chain.hip's bitrate loop
extern "C" int opusgpu_silk_nsq_del_dec_rows(const opusgpu_nsq_dd_in *d_in, opusgpu_nsq_state *d_state, opusgpu_nsq_dd_out *d_out, const int *d_rows,
                                             int m, void *d_workspace, hipStream_t stream)
{
    if (m <= 0) return m < 0 ? OPUSGPU_BAD_ARG : OPUSGPU_OK;
    int *bad = opusgpu_bad_record_counter();
    if (!bad) return OPUSGPU_ALLOC_FAIL;
    hipLaunchKernelGGL(silk_nsq_del_dec_kernel, dim3((m + 15) / 16), dim3(64), 0, stream, d_in, d_state, d_out, (dd::Scratch *)d_workspace, m, bad,
                       d_rows);
    return opusgpu_check_launch();
}

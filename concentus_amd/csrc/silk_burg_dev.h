// silk_burg_dev.h -- silk_burg_modified (opus-fix/silk/fixed/burg_modified_FIX.c:45-275) as a device function, one lane per
// call: used by the batched Burg kernel (silk_kernels.hip) and by silk_find_LPC (silk_lpc_dev.h), which calls it twice per
// frame. x is any accessor with operator[] and operator+ (the [sample][lane] LDS block on the GPU, a pointer on the host).
// Arithmetic: the reference's x86-64 build uses the 64-bit macro forms (OPUS_FAST_INT64, silk/macros.h:47-102).
#pragma once
#include "silk_math.h"

#include <utility>

namespace ca {

template <class F, int... I>
CA_DEV void ca_static_for_impl(F &f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
CA_DEV void ca_static_for(F &f) { ca_static_for_impl(f, std::make_integer_sequence<int, N>{}); }

enum { BURG_QA = 25, BURG_COND_FAC_Q32 = 42950 };                    // SILK_FIX_CONST(FIND_LPC_COND_FAC = 1e-5f, 32)

// The order recursion touches only the first and the last 16 samples of every subframe (x[n - k], x[subfr_length - n + k - 1],
// k <= n < D <= 16); the passes over whole subframes (energy, lag products) read each sample once. The two kinds of access go
// through two accessors so that a kernel can keep just the edges where latency counts (LDS) and stream the rest from where the
// signal lies: XE::head(s, i) = x[s * subfr_length + i], XE::tail(s, j) = x[s * subfr_length + subfr_length - 16 + j], i, j < 16.
#if defined(CA_HOST_EMU)
#define CA_MEMBER inline
#else
#define CA_MEMBER __device__ __forceinline__
#endif
template <class XA>
struct BurgEdgesOf {                                                  // the edges read straight from the signal
    XA x;
    int L;
    CA_MEMBER i32 head(int s, int i) const { return (i32)x[s * L + i]; }
    CA_MEMBER i32 tail(int s, int j) const { return (i32)x[s * L + L - 16 + j]; }
    CA_MEMBER BurgEdgesOf from(int s0) const { BurgEdgesOf r; r.x = x + s0 * L; r.L = L; return r; }
    template <class XB> CA_MEMBER void stage(XB, int, int) const {}
};

// the edges of up to four subframes in this lane's column of an LDS block laid out [slot][64 lanes]: 4 x 32 slots = 16 KB per
// 64-frame workgroup (the whole signal would be 48 KB, i.e. three workgroups per CU instead of four: one wave per SIMD short)
enum { BURG_EDGE_SLOTS = 4 * 32 };
struct BurgEdgesCol {
    i16 *p;
    CA_MEMBER i32 head(int s, int i) const { return (i32)p[(s * 32 + i) * 64]; }
    CA_MEMBER i32 tail(int s, int j) const { return (i32)p[(s * 32 + 16 + j) * 64]; }
    CA_MEMBER BurgEdgesCol from(int s0) const { BurgEdgesCol r; r.p = p + s0 * 32 * 64; return r; }
    template <class XB> CA_MEMBER void stage(XB x, int L, int nb) const
    {
        for (int s = 0; s < nb; s++)
            for (int i = 0; i < 16; i++) {
                p[(s * 32 + i) * 64] = (i16)(i32)x[s * L + i];
                p[(s * 32 + 16 + i) * 64] = (i16)(i32)x[s * L + L - 16 + i];
            }
    }
};

// A_Q16[16] out (entries >= D zeroed), *res_nrg / *res_nrg_Q out.
template <class XA, class XE>
CA_DEV void silk_burg_modified_dev(XA x, XE e, const i32 minInvGain_Q30, const int subfr_length, const int nb_subfr, const int D,
                                   i32 *A_Q16, i32 *res_nrg, int *res_nrg_Q)
{
    enum { QA = BURG_QA, COND_FAC_Q32 = BURG_COND_FAC_Q32 };
    i32 C_first_row[16], C_last_row[16], Af_QA[16], CAf[17], CAb[17];
    i32 C0, num, nrg, rc_Q31, invGain_Q30, Atmp_QA, Atmp1, tmp1, tmp2, x1, x2;
    int k, n, s, lz, rshifts, reached_max_gain;
    // ONE pass over the signal: its energy and the lag-1 .. lag-16 products of every subframe. The 16 previous samples travel in a
    // register window that starts at zero at each subframe (so the first n products of lag n vanish, as the reference's loop
    // bounds make them), the sums are kept exact in 64 bits per subframe until the energy has fixed rshifts: what
    // burg_modified_FIX.c:77-99 adds is then (i32)(sum >> rshifts) (its 64-bit inner products), or the low 32 bits -- the
    // wrapping sums of celt_pitch_xcorr, which do not depend on the order of the products -- shifted up.
    i64 C0_64 = 0;
    i64 acc[4][16];
#pragma unroll
    for (s = 0; s < 4; s++) {
#pragma unroll
        for (n = 0; n < 16; n++) acc[s][n] = 0;
        if (s < nb_subfr) {
            const XA xp = x + s * subfr_length;
            i32 w[16];
#pragma unroll
            for (n = 0; n < 16; n++) w[n] = 0;
#pragma unroll 4
            for (k = 0; k < subfr_length; k++) {
                const i32 xk = xp[k];
                C0_64 += __mul24(xk, xk);
#pragma unroll
                for (n = 0; n < 16; n++) acc[s][n] += __mul24(xk, w[n]);
#pragma unroll
                for (n = 15; n > 0; n--) w[n] = w[n - 1];
                w[0] = xk;
            }
        }
    }
    {
        i32 hi = (i32)(C0_64 >> 32);
        lz = hi == 0 ? 32 + s_clz32((i32)C0_64) : s_clz32(hi);
    }
    rshifts = 32 + 1 + 2 - lz;
    if (rshifts > 32 - QA) rshifts = 32 - QA;
    if (rshifts < -16) rshifts = -16;
    C0 = rshifts > 0 ? (i32)(C0_64 >> rshifts) : shl32((i32)C0_64, -rshifts);
    CAb[0] = CAf[0] = s_addw(s_addw(C0, s_smmul(COND_FAC_Q32, C0)), 1);
#pragma unroll
    for (k = 0; k < 16; k++) { C_first_row[k] = 0; Af_QA[k] = 0; }
#pragma unroll
    for (s = 0; s < 4; s++) {
        if (s < nb_subfr) {
#pragma unroll
            for (n = 0; n < 16; n++)
                if (n < D) C_first_row[n] = s_addw(C_first_row[n], rshifts > 0 ? (i32)(acc[s][n] >> rshifts) : shl32((i32)acc[s][n], -rshifts));
        }
    }
#pragma unroll
    for (k = 0; k < 16; k++) C_last_row[k] = C_first_row[k];
    invGain_Q30 = (i32)1 << 30;
    reached_max_gain = 0;
    // The order recursion, unrolled over n at compile time: with n a constant every index into C_first_row / C_last_row / Af_QA /
    // CAf / CAb below is one too, so the five arrays live in registers (indexed at run time they are private memory, and each of
    // their ~1 500 dependent updates is a round trip to it). Steps with n >= D, or after the gain limit was hit, do nothing.
    auto order_step = [&](auto NC) __attribute__((always_inline)) {
        constexpr int n = decltype(NC)::value;
        if (n >= D || reached_max_gain) return;
        i32 tmp1, tmp2, x1, x2, num, nrg, rc_Q31, Atmp_QA, Atmp1;
        int lz;
        if (rshifts > -2) {
            for (int s = 0; s < nb_subfr; s++) {
                x1 = (i32)(0u - (u32)shl32(e.head(s, n), 16 - rshifts));
                x2 = (i32)(0u - (u32)shl32(e.tail(s, 16 - n - 1), 16 - rshifts));
                tmp1 = shl32(e.head(s, n), QA - 16);
                tmp2 = shl32(e.tail(s, 16 - n - 1), QA - 16);
#pragma unroll
                for (int k = 0; k < n; k++) {
                    const i32 h = e.head(s, n - k - 1), t = e.tail(s, 16 - n + k);
                    C_first_row[k] = s_smlawb(C_first_row[k], x1, h);
                    C_last_row[k] = s_smlawb(C_last_row[k], x2, t);
                    Atmp_QA = Af_QA[k];
                    tmp1 = s_smlawb(tmp1, Atmp_QA, h);
                    tmp2 = s_smlawb(tmp2, Atmp_QA, t);
                }
                tmp1 = shl32((i32)(0u - (u32)tmp1), 32 - QA - rshifts);
                tmp2 = shl32((i32)(0u - (u32)tmp2), 32 - QA - rshifts);
#pragma unroll
                for (int k = 0; k <= n; k++) {
                    CAf[k] = s_smlawb(CAf[k], tmp1, e.head(s, n - k));
                    CAb[k] = s_smlawb(CAb[k], tmp2, e.tail(s, 16 - n + k - 1));
                }
            }
        } else {
            for (int s = 0; s < nb_subfr; s++) {
                x1 = (i32)(0u - (u32)shl32(e.head(s, n), -rshifts));
                x2 = (i32)(0u - (u32)shl32(e.tail(s, 16 - n - 1), -rshifts));
                tmp1 = shl32(e.head(s, n), 17);
                tmp2 = shl32(e.tail(s, 16 - n - 1), 17);
#pragma unroll
                for (int k = 0; k < n; k++) {
                    const i32 h = e.head(s, n - k - 1), t = e.tail(s, 16 - n + k);
                    C_first_row[k] = (i32)((u32)C_first_row[k] + (u32)x1 * (u32)h);
                    C_last_row[k] = (i32)((u32)C_last_row[k] + (u32)x2 * (u32)t);
                    Atmp1 = s_rshift_round(Af_QA[k], QA - 17);
                    tmp1 = (i32)((u32)tmp1 + (u32)h * (u32)Atmp1);
                    tmp2 = (i32)((u32)tmp2 + (u32)t * (u32)Atmp1);
                }
                tmp1 = (i32)(0u - (u32)tmp1);
                tmp2 = (i32)(0u - (u32)tmp2);
#pragma unroll
                for (int k = 0; k <= n; k++) {
                    CAf[k] = s_smlaww(CAf[k], tmp1, shl32(e.head(s, n - k), -rshifts - 1));
                    CAb[k] = s_smlaww(CAb[k], tmp2, shl32(e.tail(s, 16 - n + k - 1), -rshifts - 1));
                }
            }
        }
        tmp1 = C_first_row[n];
        tmp2 = C_last_row[n];
        num = 0;
        nrg = s_addw(CAb[0], CAf[0]);
#pragma unroll
        for (int k = 0; k < n; k++) {
            Atmp_QA = Af_QA[k];
            lz = s_clz32(s_abs(Atmp_QA)) - 1;
            if (lz > 32 - QA) lz = 32 - QA;
            Atmp1 = shl32(Atmp_QA, lz);
            tmp1 = s_addw(tmp1, shl32(s_smmul(C_last_row[n - k - 1], Atmp1), 32 - QA - lz));
            tmp2 = s_addw(tmp2, shl32(s_smmul(C_first_row[n - k - 1], Atmp1), 32 - QA - lz));
            num = s_addw(num, shl32(s_smmul(CAb[n - k], Atmp1), 32 - QA - lz));
            nrg = s_addw(nrg, shl32(s_smmul(s_addw(CAb[k + 1], CAf[k + 1]), Atmp1), 32 - QA - lz));
        }
        CAf[n + 1] = tmp1;
        CAb[n + 1] = tmp2;
        num = s_addw(num, tmp2);
        num = shl32((i32)(0u - (u32)num), 1);
        if (s_abs(num) < nrg) rc_Q31 = s_div32_varq(num, nrg, 31);
        else rc_Q31 = (num > 0) ? 0x7FFFFFFF : (i32)0x80000000;
        tmp1 = s_subw((i32)1 << 30, s_smmul(rc_Q31, rc_Q31));
        tmp1 = shl32(s_smmul(invGain_Q30, tmp1), 2);
        if (tmp1 <= minInvGain_Q30) {
            tmp2 = s_subw((i32)1 << 30, s_div32_varq(minInvGain_Q30, invGain_Q30, 30));
            rc_Q31 = s_sqrt_approx(tmp2);
            rc_Q31 = s_addw(rc_Q31, tmp2 / rc_Q31) >> 1;
            rc_Q31 = shl32(rc_Q31, 16);
            if (num < 0) rc_Q31 = (i32)(0u - (u32)rc_Q31);
            invGain_Q30 = minInvGain_Q30;
            reached_max_gain = 1;
        } else {
            invGain_Q30 = tmp1;
        }
#pragma unroll
        for (int k = 0; k < (n + 1) >> 1; k++) {
            tmp1 = Af_QA[k];
            tmp2 = Af_QA[n - k - 1];
            Af_QA[k] = s_addw(tmp1, shl32(s_smmul(tmp2, rc_Q31), 1));
            Af_QA[n - k - 1] = s_addw(tmp2, shl32(s_smmul(tmp1, rc_Q31), 1));
        }
        Af_QA[n] = rc_Q31 >> (31 - QA);
        if (reached_max_gain) {
#pragma unroll
            for (int k = n + 1; k < 16; k++) if (k < D) Af_QA[k] = 0;
            return;
        }
#pragma unroll
        for (int k = 0; k <= n + 1; k++) {
            tmp1 = CAf[k];
            tmp2 = CAb[n - k + 1];
            CAf[k] = s_addw(tmp1, shl32(s_smmul(tmp2, rc_Q31), 1));
            CAb[n - k + 1] = s_addw(tmp2, shl32(s_smmul(tmp1, rc_Q31), 1));
        }
    };
    ca_static_for<16>(order_step);
    if (reached_max_gain) {
#pragma unroll
        for (k = 0; k < 16; k++) if (k < D) A_Q16[k] = (i32)(0u - (u32)s_rshift_round(Af_QA[k], QA - 16));
        if (rshifts > 0) {
            for (s = 0; s < nb_subfr; s++) {
                i64 acc = 0;
                for (k = 0; k < D; k++) acc += __mul24(e.head(s, k), e.head(s, k));
                C0 = s_subw(C0, (i32)(acc >> rshifts));
            }
        } else {
            for (s = 0; s < nb_subfr; s++) {
                i32 acc = 0;
                for (k = 0; k < D; k++) acc = s_addw(acc, __mul24(e.head(s, k), e.head(s, k)));
                C0 = s_subw(C0, shl32(acc, -rshifts));
            }
        }
        *res_nrg = shl32(s_smmul(invGain_Q30, C0), 2);
        *res_nrg_Q = -rshifts;
    } else {
        nrg = CAf[0];
        tmp1 = (i32)1 << 16;
#pragma unroll
        for (k = 0; k < 16; k++) {
            if (k < D) {
                Atmp1 = s_rshift_round(Af_QA[k], QA - 16);
                nrg = s_smlaww(nrg, CAf[k + 1], Atmp1);
                tmp1 = s_smlaww(tmp1, Atmp1, Atmp1);
                A_Q16[k] = (i32)(0u - (u32)Atmp1);
            }
        }
        *res_nrg = s_smlaww(nrg, s_smmul(COND_FAC_Q32, C0), (i32)(0u - (u32)tmp1));
        *res_nrg_Q = -rshifts;
    }
    for (k = D; k < 16; k++) A_Q16[k] = 0;
}

template <class XA>
CA_DEV void silk_burg_modified_dev(XA x, const i32 minInvGain_Q30, const int subfr_length, const int nb_subfr, const int D,
                                   i32 *A_Q16, i32 *res_nrg, int *res_nrg_Q)
{
    BurgEdgesOf<XA> e;
    e.x = x; e.L = subfr_length;
    silk_burg_modified_dev(x, e, minInvGain_Q30, subfr_length, nb_subfr, D, A_Q16, res_nrg, res_nrg_Q);
}

}  // namespace ca


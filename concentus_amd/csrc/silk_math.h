// silk_math.h -- SILK fixed-point macro forms shared by the SILK kernels (opus-fix/silk/macros.h:47-102 in the
// 64-bit forms of the x86-64 reference build, silk/SigProc_FIX.h, silk/Inlines.h:68-186). 16x32 products are evaluated
// on split halves (full-rate 24-bit multiplier), 32x32 ones as 64-bit.
#pragma once
#include "fixmath.h"

namespace ca {

CA_DEV i32 s_smulwb(i32 a, i32 b16) { b16 = (i16)b16; return (i32)((u32)__mul24(a >> 16, b16) + (u32)(__mul24((i32)(a & 0xffff), b16) >> 16)); }
CA_DEV i32 s_smlawb(i32 a, i32 b, i32 c16) { return (i32)((u32)a + (u32)s_smulwb(b, c16)); }
CA_DEV i32 s_smlawt(i32 a, i32 b, i32 c) { return (i32)((u32)a + (u32)s_smulwb(b, c >> 16)); }
CA_DEV i32 s_smulww(i32 a, i32 b) { return (i32)(((i64)a * b) >> 16); }
CA_DEV i32 s_smlaww(i32 a, i32 b, i32 c) { return (i32)((u32)a + (u32)s_smulww(b, c)); }
CA_DEV i32 s_smulbb(i32 a, i32 b) { return __mul24((i32)(i16)a, (i32)(i16)b); }
CA_DEV i32 s_smmul(i32 a, i32 b) { return (i32)(((i64)a * b) >> 32); }
CA_DEV i32 s_rshift_round(i32 a, int s) { return s == 1 ? (a >> 1) + (a & 1) : ((a >> (s - 1)) + 1) >> 1; }
CA_DEV i32 s_abs(i32 a) { return a > 0 ? a : (i32)(0u - (u32)a); }
CA_DEV int s_clz32(i32 x) { return x ? __clz(x) : 32; }
CA_DEV i32 s_limit(i32 a, i32 l1, i32 l2) { return l1 > l2 ? (a > l1 ? l1 : (a < l2 ? l2 : a)) : (a > l2 ? l2 : (a < l1 ? l1 : a)); }
CA_DEV i32 s_lshift_sat32(i32 a, int s) { return shl32(s_limit(a, (i32)0x80000000 >> s, 0x7FFFFFFF >> s), s); }
CA_DEV i32 s_addw(i32 a, i32 b) { return (i32)((u32)a + (u32)b); }
CA_DEV i32 s_subw(i32 a, i32 b) { return (i32)((u32)a - (u32)b); }

// Eight consecutive 16-bit samples as ONE 16-byte access (no alignment assumed). A lane that streams through its own record
// touches a cache line of its own per instruction, so what a signal in a record costs is the number of memory instructions, not
// the bytes: the loops over the signals of a record (silk_pitch_dev.h, silk_pred_kernels.hip) move eight samples per instruction.
struct Pack8 { i16 s[8]; };
template <class P>
CA_DEV void pe_load8(i32 *v, P p)
{
    Pack8 t;
    __builtin_memcpy(&t, &p[0], sizeof(t));
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = t.s[u];
}
template <class P>
CA_DEV void pe_store8(P p, const i32 *v)
{
    Pack8 t;
#pragma unroll
    for (int u = 0; u < 8; u++) t.s[u] = (i16)v[u];
    __builtin_memcpy(&p[0], &t, sizeof(t));
}

// a / b for 0 <= a < 2^22, 0 < b < 2^22, exact: the quotient estimated through a single-precision reciprocal is at most one
// off, one comparison of the remainder settles it (a generic 32-bit signed division is ~40 instructions on the device)
CA_DEV i32 s_div_small(i32 a, i32 b)
{
#if defined(CA_HOST_EMU)
    const float rb = 1.0f / (float)b;
#else
    const float rb = __builtin_amdgcn_rcpf((float)b);
#endif
    i32 q = (i32)((float)a * rb);
    const i32 r = a - __mul24(q, b);
    q += r < 0 ? -1 : r >= b ? 1 : 0;
    return q;
}

CA_DEV i32 s_div32_varq(i32 a32, i32 b32, int Qres)                 // Inlines.h:96-139
{
    int a_headrm = s_clz32(s_abs(a32)) - 1;
    i32 a32_nrm = shl32(a32, a_headrm);
    int b_headrm = s_clz32(s_abs(b32)) - 1;
    i32 b32_nrm = shl32(b32, b_headrm);
    i32 b32_inv = (0x7FFFFFFF >> 2) / (b32_nrm >> 16);
    i32 result = s_smulwb(a32_nrm, b32_inv);
    a32_nrm = (i32)((u32)a32_nrm - ((u32)s_smmul(b32_nrm, result) << 3));
    result = s_smlawb(result, a32_nrm, b32_inv);
    int lshift = 29 + a_headrm - b_headrm - Qres;
    if (lshift < 0) return s_lshift_sat32(result, -lshift);
    return lshift < 32 ? result >> lshift : 0;
}

CA_DEV i32 s_inverse32_varq(i32 b32, int Qres)                      // Inlines.h:142-186
{
    int b_headrm = s_clz32(s_abs(b32)) - 1;
    i32 b32_nrm = shl32(b32, b_headrm);
    i32 b32_inv = (0x7FFFFFFF >> 2) / (b32_nrm >> 16);
    i32 result = shl32(b32_inv, 16);
    i32 err_Q32 = shl32(((i32)1 << 29) - s_smulwb(b32_nrm, b32_inv), 3);
    result = s_smlaww(result, err_Q32, b32_inv);
    int lshift = 61 - b_headrm - Qres;
    if (lshift <= 0) return s_lshift_sat32(result, -lshift);
    return lshift < 32 ? result >> lshift : 0;
}

CA_DEV i32 s_sqrt_approx(i32 x)                                     // Inlines.h:68-93
{
    if (x <= 0) return 0;
    int lz = s_clz32(x);
    int rot = 24 - lz;
    u32 ux = (u32)x;
    u32 rr = rot == 0 ? ux : rot < 0 ? ((ux << (u32)-rot) | (ux >> (32 - (u32)-rot))) : ((ux << (32 - rot)) | (ux >> rot));
    i32 frac_Q7 = (i32)(rr & 0x7f);
    i32 y = (lz & 1) ? 32768 : 46214;
    y >>= (lz >> 1);
    return s_smlawb(y, y, s_smulbb(213, frac_Q7));
}

}  // namespace ca

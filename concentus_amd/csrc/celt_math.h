// celt_math.h -- bit-exact counterparts of the reference's fixed-point math approximations
// (opus-fix/celt/mathops.c:42-206, celt/mathops.h:40-258, celt/bands.c:66-93). Every intermediate
// that the reference keeps in an opus_val16 is narrowed with (i16) at the same point; SUB16 does NOT
// narrow its result in the reference (fixed_generic.h:91) and does not here either.
#pragma once
#include "fixmath.h"

namespace ca {

// ecintrin.h: EC_ILOG(x) = 32 - clz(x) for x > 0 (0 for x == 0)
CA_DEV int ec_ilog(u32 x) { return 32 - __clz((int)x); }
CA_DEV int celt_ilog2(i32 x) { return ec_ilog((u32)x) - 1; }                 // mathops.h:178
CA_DEV int celt_zlog2(i32 x) { return x <= 0 ? 0 : celt_ilog2(x); }            // mathops.h:184

CA_DEV i32 add16(i32 a, i32 b) { return (i16)((i16)a + (i16)b); }              // ADD16 narrows
CA_DEV i32 sub16(i32 a, i32 b) { return (i32)(i16)a - (i32)(i16)b; }           // SUB16 does not
CA_DEV i32 shl16(i32 a, int s) { return (i16)((u16)a << s); }
CA_DEV i32 frac_mul16(i32 a, i32 b) { return (16384 + (i32)(i16)a * (i32)(i16)b) >> 15; }   // mathops.h:44

CA_DEV u32 isqrt32(u32 val)                                                    // mathops.c:42-67
{
    u32 g = 0;
    int bshift = (ec_ilog(val) - 1) >> 1;
    u32 b = 1u << bshift;
    do {
        u32 t = ((g << 1) + b) << bshift;
        if (t <= val) { g += b; val -= t; }
        b >>= 1;
        bshift--;
    } while (bshift >= 0);
    return g;
}

CA_DEV i32 celt_rcp(i32 x)                                                     // mathops.c:180-206
{
    int i = celt_ilog2(x);
    i32 n = (i16)(vshr32(x, i - 15) - 32768);
    i32 r = add16(30840, mul16_16_q15(-15420, n));
    r = (i16)sub16(r, mul16_16_q15(r, add16(mul16_16_q15(r, n), add16(r, -32768))));
    r = (i16)sub16(r, add16(1, mul16_16_q15(r, add16(mul16_16_q15(r, n), add16(r, -32768)))));
    return vshr32(r, i - 16);
}

CA_DEV i32 celt_div(i32 a, i32 b) { return mul32_32_q31(a, celt_rcp(b)); }     // mathops.h:226

CA_DEV i32 frac_div32(i32 a, i32 b)                                            // mathops.c:69-89
{
    int shift = celt_ilog2(b) - 29;
    a = vshr32(a, shift);
    b = vshr32(b, shift);
    i32 rcp = (i16)pshr32(celt_rcp((i16)pshr32(b, 16)), 3);
    i32 result = mul16_32_q15(rcp, a);
    i32 rem = sub32(pshr32(a, 2), mul32_32_q31(result, b));
    result = add32(result, shl32(mul16_32_q15(rcp, rem), 2));
    if (result >= 536870912) return 2147483647;
    if (result <= -536870912) return -2147483647;
    return shl32(result, 2);
}

CA_DEV i32 celt_rsqrt_norm(i32 x)                                              // mathops.c:92-114
{
    i32 n = (i16)(x - 32768);
    i32 r = add16(23557, mul16_16_q15(n, add16(-13490, mul16_16_q15(n, 6713))));
    i32 r2 = (i16)mul16_16_q15(r, r);
    i32 y = shl16(sub16(add16(mul16_16_q15(r2, n), r2), 16384), 1);
    return add16(r, mul16_16_q15(r, mul16_16_q15(y, sub16(mul16_16_q15(y, 12288), 16384))));
}

CA_DEV i32 celt_sqrt(i32 x)                                                    // mathops.c:117-135
{
    if (x == 0) return 0;
    if (x >= 1073741824) return 32767;
    int k = (celt_ilog2(x) >> 1) - 7;
    x = vshr32(x, 2 * k);
    i32 n = (i16)(x - 32768);
    i32 rt = add16(23175, mul16_16_q15(n, add16(11561, mul16_16_q15(n, add16(-3011,
                 mul16_16_q15(n, add16(1699, mul16_16_q15(n, -664))))))));
    return vshr32(rt, 7 - k);
}

CA_DEV i32 celt_cos_pi_2(i32 x)                                                // mathops.c:142-150
{
    i32 x2 = (i16)mul16_16_p15(x, x);
    i32 t = add32(8277, mul16_16_p15(-626, x2));
    t = add32(-7651, mul16_16_p15(x2, t));
    t = add32(sub16(32767, x2), mul16_16_p15(x2, t));
    return add16(1, imin(32766, t));
}

CA_DEV i32 celt_cos_norm(i32 x)                                                // mathops.c:156-177
{
    x &= 0x0001ffff;
    if (x > (1 << 16)) x = (1 << 17) - x;
    if (x & 0x00007fff) {
        if (x < (1 << 15)) return celt_cos_pi_2((i16)x);
        return (i16)neg32(celt_cos_pi_2((i16)(65536 - x)));
    }
    if (x & 0x0000ffff) return 0;
    if (x & 0x0001ffff) return -32767;
    return 32767;
}

CA_DEV i32 celt_log2(i32 x)                                                    // mathops.h:192-204 (DB_SHIFT 10)
{
    if (x == 0) return -32767;
    int i = celt_ilog2(x);
    i32 n = (i16)(vshr32(x, i - 15) - 32768 - 16384);
    i32 frac = add16(-6793, mul16_16_q15(n, add16(15746, mul16_16_q15(n, add16(-5217,
                   mul16_16_q15(n, add16(2545, mul16_16_q15(n, -1401))))))));
    return (i16)(shl16(i - 13, 10) + (frac >> 4));
}

CA_DEV i32 celt_exp2_frac(i32 x)                                               // mathops.h:214-219
{
    i32 frac = shl16(x, 4);
    return add16(16383, mul16_16_q15(frac, add16(22804, mul16_16_q15(frac, add16(14819, mul16_16_q15(10204, frac))))));
}

CA_DEV i32 celt_exp2(i32 x)                                                    // mathops.h:221-232
{
    int integer = (i16)x >> 10;
    if (integer > 14) return 0x7f000000;
    if (integer < -15) return 0;
    i32 frac = celt_exp2_frac((i16)(x - shl16(integer, 10)));
    return vshr32(frac, -integer - 2);
}

CA_DEV i32 celt_atan01(i32 x)                                                  // mathops.h:236-239
{
    return (i16)mul16_16_p15(x, add32(32767, mul16_16_p15(x, add32(-21, mul16_16_p15(x, add32(-11943, mul16_16_p15(4936, x)))))));
}

CA_DEV i32 celt_atan2p(i32 y, i32 x)                                           // mathops.h:246-261
{
    if (y < x) {
        i32 arg = celt_div(shl32(y, 15), x);
        if (arg >= 32767) arg = 32767;
        return celt_atan01((i16)arg) >> 1;
    }
    i32 arg = celt_div(shl32(x, 15), y);
    if (arg >= 32767) arg = 32767;
    return (i16)(25736 - (celt_atan01((i16)arg) >> 1));
}

CA_DEV i32 bitexact_cos(i32 x)                                                 // bands.c:66-77
{
    i32 tmp = (4096 + x * x) >> 13;
    i32 x2 = (i16)tmp;
    x2 = (i16)((32767 - x2) + frac_mul16(x2, (-7651 + frac_mul16(x2, (8277 + frac_mul16(-626, x2))))));
    return (i16)(1 + x2);
}

CA_DEV i32 bitexact_log2tan(i32 isin, i32 icos)                                // bands.c:79-93
{
    int lc = ec_ilog((u32)icos), ls = ec_ilog((u32)isin);
    icos <<= 15 - lc;
    isin <<= 15 - ls;
    return (ls - lc) * (1 << 11) + frac_mul16(isin, frac_mul16(isin, -2597) + 7932)
         - frac_mul16(icos, frac_mul16(icos, -2597) + 7932);
}

}  // namespace ca

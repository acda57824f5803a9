// fixmath.h -- device-side L0 fixed-point arithmetic for gfx950.
//
// Bit-exact counterparts of the reference's Q-format macros (opus-fix/celt/fixed_generic.h:36-151).
// Everything is written on the split 16-bit halves so the products fit the full-rate 24-bit
// integer multiplier (v_mul_i32_i24 / v_mad_i32_i24) instead of the quarter-rate 32x32 multiplier;
// wrap-around on overflow is two's complement, as in the reference binary.
#pragma once
#include "wave.h"

namespace ca {

typedef int16_t i16;
typedef uint16_t u16;
typedef uint8_t u8;
typedef int8_t i8;
typedef int32_t i32;
typedef uint32_t u32;
typedef int64_t i64;
typedef uint64_t u64;

// fixed_generic.h:46 MULT16_32_Q15(a,b) = (a*(b>>16))<<1 + (a*(b&0xffff))>>15
CA_DEV i32 mul16_32_q15(i32 a16, i32 b)
{
    i32 hi = __mul24(a16, b >> 16);
    i32 lo = __mul24(a16, (i32)(b & 0xffff)) >> 15;
    return (i32)(((u32)hi << 1) + (u32)lo);
}

// fixed_generic.h:40 MULT16_32_Q16
CA_DEV i32 mul16_32_q16(i32 a16, i32 b)
{
    return (i32)((u32)__mul24(a16, b >> 16) + (u32)(__mul24(a16, (i32)(b & 0xffff)) >> 16));
}

// fixed_generic.h:43 MULT16_32_P16
CA_DEV i32 mul16_32_p16(i32 a16, i32 b)
{
    return (i32)((u32)__mul24(a16, b >> 16) + (u32)((__mul24(a16, (i32)(b & 0xffff)) + 32768) >> 16));
}

// fixed_generic.h:49 MULT32_32_Q31
CA_DEV i32 mul32_32_q31(i32 a, i32 b)
{
    i32 ah = a >> 16, bh = b >> 16;
    u32 t0 = (u32)__mul24(ah, bh) << 1;
    i32 t1 = __mul24(ah, (i32)(b & 0xffff)) >> 15;
    i32 t2 = __mul24(bh, (i32)(a & 0xffff)) >> 15;
    return (i32)(t0 + (u32)t1 + (u32)t2);
}

// fixed_generic.h:116 MAC16_32_Q15 (splits b at bit 15)
CA_DEV i32 mac16_32_q15(i32 c, i32 a16, i32 b)
{
    return (i32)((u32)c + (u32)__mul24(a16, b >> 15) + (u32)(__mul24(a16, (i32)(b & 0x7fff)) >> 15));
}

CA_DEV i32 mul16_16(i32 a, i32 b) { return __mul24((i32)(i16)a, (i32)(i16)b); }
CA_DEV i32 mac16_16(i32 c, i32 a, i32 b) { return (i32)((u32)c + (u32)mul16_16(a, b)); }
CA_DEV i32 mul16_16_q15(i32 a, i32 b) { return mul16_16(a, b) >> 15; }
CA_DEV i32 mul16_16_p15(i32 a, i32 b) { return (16384 + mul16_16(a, b)) >> 15; }
CA_DEV i32 mul16_16_q14(i32 a, i32 b) { return mul16_16(a, b) >> 14; }
CA_DEV i32 mul16_16_q13(i32 a, i32 b) { return mul16_16(a, b) >> 13; }
CA_DEV i32 mul16_16_q11(i32 a, i32 b) { return mul16_16(a, b) >> 11; }
CA_DEV i32 mul16_16_p13(i32 a, i32 b) { return (4096 + mul16_16(a, b)) >> 13; }
CA_DEV i32 mul16_16_p14(i32 a, i32 b) { return (8192 + mul16_16(a, b)) >> 14; }

CA_DEV i32 shl32(i32 a, int s) { return (i32)((u32)a << s); }
CA_DEV i32 pshr32(i32 a, int s) { return (i32)((u32)a + (u32)((1 << s) >> 1)) >> s; }
CA_DEV i32 vshr32(i32 a, int s) { return s > 0 ? a >> s : shl32(a, -s); }
CA_DEV i32 add32(i32 a, i32 b) { return (i32)((u32)a + (u32)b); }
CA_DEV i32 sub32(i32 a, i32 b) { return (i32)((u32)a - (u32)b); }
CA_DEV i32 neg32(i32 a) { return (i32)(0u - (u32)a); }
CA_DEV i32 imin(i32 a, i32 b) { return a < b ? a : b; }
CA_DEV i32 imax(i32 a, i32 b) { return a > b ? a : b; }

}  // namespace ca

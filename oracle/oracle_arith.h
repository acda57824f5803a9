/* oracle/oracle_arith.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's L0 fixed-point arithmetic (opus-fix/celt/fixed_generic.h:36-151,
 * celt/arch.h:83-111). Plain C, compiled with -fwrapv so signed overflow wraps exactly as it does
 * in the reference binary. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * use anything under oracle/.
 */
#ifndef ORACLE_ARITH_H
#define ORACLE_ARITH_H
#include <stdint.h>

typedef int16_t i16;
typedef int32_t i32;
typedef uint32_t u32;
typedef int64_t i64;
typedef int8_t i8;

/* fixed_generic.h:46  MULT16_32_Q15(a,b): 16x32 -> >>15, evaluated on the split halves of b.
 * The split form equals ((int64)a*b)>>15 truncated to 32 bits; it is written the reference's way so
 * the wrap-around behaviour is visibly the same. */
static inline i32 mul16_32_q15(i16 a, i32 b)
{
    i32 hi = (i32)a * (b >> 16);
    i32 lo = ((i32)a * (i32)(b & 0xffff)) >> 15;
    return (i32)((u32)hi << 1) + lo;
}

/* fixed_generic.h:40  MULT16_32_Q16 */
static inline i32 mul16_32_q16(i16 a, i32 b)
{
    return (i32)a * (b >> 16) + (((i32)a * (i32)(b & 0xffff)) >> 16);
}

/* fixed_generic.h:43  MULT16_32_P16 (rounded low half) */
static inline i32 mul16_32_p16(i16 a, i32 b)
{
    return (i32)a * (b >> 16) + ((((i32)a * (i32)(b & 0xffff)) + 32768) >> 16);
}

/* fixed_generic.h:49  MULT32_32_Q31 */
static inline i32 mul32_32_q31(i32 a, i32 b)
{
    i32 t0 = (i32)((u32)((i32)(i16)(a >> 16) * (i32)(i16)(b >> 16)) << 1);
    i32 t1 = ((i32)(i16)(a >> 16) * (i32)(b & 0xffff)) >> 15;
    i32 t2 = ((i32)(i16)(b >> 16) * (i32)(a & 0xffff)) >> 15;
    return t0 + t1 + t2;
}

/* fixed_generic.h:116  MAC16_32_Q15: NOTE splits b at bit 15, not 16 */
static inline i32 mac16_32_q15(i32 c, i16 a, i32 b)
{
    return c + ((i32)a * (b >> 15) + (((i32)a * (i32)(b & 0x7fff)) >> 15));
}

static inline i32 mul16_16(i32 a, i32 b) { return (i32)(i16)a * (i32)(i16)b; }
static inline i32 mac16_16(i32 c, i32 a, i32 b) { return c + mul16_16(a, b); }
static inline i32 mul16_16_q15(i32 a, i32 b) { return mul16_16(a, b) >> 15; }
static inline i32 mul16_16_p15(i32 a, i32 b) { return (16384 + mul16_16(a, b)) >> 15; }
static inline i32 mul16_16_q14(i32 a, i32 b) { return mul16_16(a, b) >> 14; }
static inline i32 mul16_16_q13(i32 a, i32 b) { return mul16_16(a, b) >> 13; }
static inline i32 mul16_16_q11(i32 a, i32 b) { return mul16_16(a, b) >> 11; }
static inline i32 mul16_16_p13(i32 a, i32 b) { return (4096 + mul16_16(a, b)) >> 13; }
static inline i32 mul16_16_p14(i32 a, i32 b) { return (8192 + mul16_16(a, b)) >> 14; }

static inline i32 shl32(i32 a, int s) { return (i32)((u32)a << s); }
static inline i32 shr32(i32 a, int s) { return a >> s; }
static inline i32 pshr32(i32 a, int s) { return (a + ((1 << s) >> 1)) >> s; }
static inline i32 vshr32(i32 a, int s) { return s > 0 ? a >> s : shl32(a, -s); }
static inline i16 round16(i32 a, int s) { return (i16)pshr32(a, s); }
static inline i16 extract16(i32 a) { return (i16)a; }
static inline i32 imin(i32 a, i32 b) { return a < b ? a : b; }
static inline i32 imax(i32 a, i32 b) { return a > b ? a : b; }
static inline i32 iabs(i32 a) { return a < 0 ? -a : a; }

#define SIG_SHIFT 12

#endif

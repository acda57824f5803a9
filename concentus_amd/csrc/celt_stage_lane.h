// celt_stage_lane.h -- the strictly serial per-channel stages of the front phase, written for ONE LANE
// per (frame, channel): dc_reject (opus-fix/src/opus_encoder.c:362-384) and the per-channel part of
// transient_analysis (opus-fix/celt/celt_encoder.c:227-377).
//
// Both are first/second-order recurrences with rounding in the loop, so they cannot be turned into
// scans; inside a one-wavefront-per-frame kernel they keep 2 of 64 lanes busy. Here 64 (frame, channel)
// pairs share a wavefront, the working set of a lane is a handful of registers and the arrays stream
// through global memory 16 bytes at a time.
#pragma once
#include "celt_math.h"
#include "device_tables.h"

namespace ca {

enum { STG_FRAME = 960, STG_OVL = 120 };

// dc_reject of channel c: pcm interleaved int16 [960][2] (16-byte aligned) -> out planar int16 [960];
// hp[2] is the filter memory (in/out). Stereo only.
CA_DEV void stage_dc_reject_channel(const i16 *__restrict__ pcm, int c, i32 *hp, i16 *__restrict__ out)
{
    i32 m0 = hp[0], m1 = hp[1];
    const int4 *src = reinterpret_cast<const int4 *>(pcm);
    int4 *dst = reinterpret_cast<int4 *>(out);
    // Blocks of 64 samples: the block's sixteen 16-byte loads are issued together, the recurrence runs over them, its eight
    // 16-byte results are stored together. (One group of eight samples per trip -- two loads, the recurrence, one store --
    // was one exposed memory round trip per trip, the loads queued behind the previous trip's store: 120 of them per
    // channel, nine tenths of the kernel's time.)
    enum { BLK = 64 };
    static_assert(STG_FRAME % BLK == 0, "dc_reject block");
    for (int i0 = 0; i0 < STG_FRAME; i0 += BLK) {
        int4 in[BLK / 4];
#pragma unroll
        for (int q = 0; q < BLK / 4; q++) in[q] = src[(i0 >> 2) + q];
        int4 res[BLK / 8];
#pragma unroll
        for (int g = 0; g < BLK / 8; g++) {
            const int4 a = in[2 * g], b = in[2 * g + 1];                   // 8 stereo pairs
            const i32 w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            u32 o16[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                i32 s = c ? (w[k] >> 16) : (i32)(i16)w[k];
                i32 x = shl32(s, 15);
                i32 tmp = sub32(x, m0);
                m0 = add32(m0, pshr32(sub32(x, m0), 12));
                i32 y = sub32(tmp, m1);
                m1 = add32(m1, pshr32(sub32(tmp, m1), 12));
                i32 o = pshr32(y, 15);
                o = o > 32767 ? 32767 : (o < -32767 ? -32767 : o);                  // SATURATE(x, 32767)
                o16[k] = (u32)o & 0xffffu;
            }
            res[g].x = (i32)(o16[0] | (o16[1] << 16)); res[g].y = (i32)(o16[2] | (o16[3] << 16));
            res[g].z = (i32)(o16[4] | (o16[5] << 16)); res[g].w = (i32)(o16[6] | (o16[7] << 16));
        }
#pragma unroll
        for (int g = 0; g < BLK / 8; g++) dst[(i0 >> 3) + g] = res[g];
    }
    hp[0] = m0;
    hp[1] = m1;
}

// One step of the transient high-pass (celt_encoder.c:262-280): returns tmp[i] before the <<shift
CA_DEV i32 stage_trans_hp(i32 in_q12, i32 &mem0, i32 &mem1)
{
    i32 x = in_q12 >> 12;
    i32 y = add32(mem0, x);
    mem0 = sub32(add32(mem1, y), shl32(x, 1));
    mem1 = sub32(x, y >> 1);
    return (i16)(y >> 2);
}

// Masking metric of one channel: in = the [1080] int32 pre-filtered time signal of the channel
// (16-byte aligned), tmp2 = 544 int16 of scratch (16-byte aligned). Returns `unmask` of celt_encoder.c:352.
// The high-pass output is recomputed in the second pass instead of being stored: the normalising shift
// depends on the maximum over the whole frame.
CA_DEV i32 stage_transient_channel(const i32 *__restrict__ in, i16 *__restrict__ tmp2)
{
    const int len = STG_FRAME + STG_OVL, len2 = len / 2;
    const int4 *src = reinterpret_cast<const int4 *>(in);
    i32 mem0 = 0, mem1 = 0, mx = 0, mn = 0;
    // the loads of several groups are issued ahead of the recurrence that consumes them
#pragma unroll 6
    for (int i0 = 0; i0 < len; i0 += 4) {
        const int4 a = src[i0 >> 2];
        const i32 w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            i32 t = stage_trans_hp(w[k], mem0, mem1);
            if (i0 + k < 12) t = 0;
            mx = imax(mx, t);
            mn = imin(mn, t);
        }
    }
    const int shift = 14 - celt_ilog2(1 + imax(mx, -mn));
    // forward follower over pair energies
    i32 mean = 0, fm = 0;
    mem0 = mem1 = 0;
#pragma unroll 3
    for (int i0 = 0; i0 < len; i0 += 8) {
        const int4 a = src[(i0 >> 2) + 0], b = src[(i0 >> 2) + 1];
        const i32 w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        i32 t[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            t[k] = stage_trans_hp(w[k], mem0, mem1);
            if (i0 + k < 12) t[k] = 0;
            if (shift != 0) t[k] = (i16)shl16(t[k], shift);
        }
        u32 f[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            i32 x2 = (i16)pshr32(add32(mul16_16(t[2 * k], t[2 * k]), mul16_16(t[2 * k + 1], t[2 * k + 1])), 16);
            mean = add32(mean, x2);
            fm = (i16)(fm + pshr32(x2 - fm, 4));
            f[k] = (u32)fm & 0xffffu;
        }
        int2 r;
        r.x = (i32)(f[0] | (f[1] << 16));
        r.y = (i32)(f[2] | (f[3] << 16));
        reinterpret_cast<int2 *>(tmp2)[i0 >> 3] = r;
    }
    // backward follower, in place; len2 = 540 = 135 groups of 4
    i32 bm = 0, maxE = 0;
#pragma unroll 5
    for (int g = len2 / 4 - 1; g >= 0; g--) {
        int2 v = reinterpret_cast<const int2 *>(tmp2)[g];
        i32 e[4] = {(i32)(i16)v.x, v.x >> 16, (i32)(i16)v.y, v.y >> 16};
        u32 f[4];
#pragma unroll
        for (int k = 3; k >= 0; k--) {
            bm = (i16)(bm + pshr32(e[k] - bm, 3));
            maxE = imax(maxE, bm);
            f[k] = (u32)bm & 0xffffu;
        }
        int2 r;
        r.x = (i32)(f[0] | (f[1] << 16));
        r.y = (i32)(f[2] | (f[3] << 16));
        reinterpret_cast<int2 *>(tmp2)[g] = r;
    }
    mean = mul16_16(celt_sqrt(mean), celt_sqrt(mul16_16(maxE, len2 >> 1)));
    const i32 norm = shl32(len2, 6 + 14) / add32(1, mean >> 1);
    i32 unmask = 0;
    for (int i = 12; i < len2 - 5; i += 4) {
        i32 id = imax(0, imin(127, mul16_32_q15((i16)(tmp2[i] + 1), norm)));
        unmask += CLT_inv_table[id];
    }
    return 64 * unmask * 4 / (6 * (len2 - 17));
}

}  // namespace ca

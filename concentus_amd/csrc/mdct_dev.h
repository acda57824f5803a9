// mdct_dev.h -- wave-cooperative CELT MDCT / FFT for gfx950 (one 64-lane wavefront per transform).
//
// Device-side building blocks shared by the MDCT-only kernels (mdct_kernels.hip) and the full CELT
// frame kernels. All working arrays live in LDS and belong to ONE wavefront; `wave_sync()` orders the
// LDS traffic between its lanes. The arithmetic per element follows the reference bit for bit
// (opus-fix/celt/kiss_fft.c:51-322,532-578 and celt/mdct.c:121-363); what is new is the mapping:
// every butterfly stage is one flat index space of 480/p butterflies spread over the 64 lanes, and
// the 8 short blocks of a transient frame run side by side in the same index space.
#pragma once
#include "fixmath.h"
#include "device_tables.h"

namespace ca {

// ---- constant tables --------------------------------------------------------------------------------
// The transforms read their tables through an MdctTab of pointers, so a kernel may point it at the
// device-global packed tables (celt_tables.h) or at copies it staged into LDS.
//   tw      fft twiddles, packed (r & 0xffff) | (i << 16)              static_modes_fixed.h:104
//   trig    packed (trig[i] & 0xffff) | (trig[N4+i] << 16) for the shift in use (mdct.c:141-146)
//   bitrev  fft_bitrev{480,240,120,60} for the shift in use
struct MdctTab {
    const u32 *tw;
    const u32 *trig;
    const i16 *bitrev;
    const i16 *window;
    const i16 *bitrev_sw;          // shift 0: fsw<0>(bitrev[i]), where the pre-rotation of the long transform stores input i
};

template <int SHIFT>
CA_DEV MdctTab mdct_global_tab()
{
    MdctTab t;
    t.tw = CLT_fft_tw_packed;
    t.trig = SHIFT == 0 ? CLT_mdct_trig_packed0 : SHIFT == 1 ? CLT_mdct_trig_packed1
           : SHIFT == 2 ? CLT_mdct_trig_packed2 : CLT_mdct_trig_packed3;
    t.bitrev = SHIFT == 0 ? CLT_fft_bitrev480 : SHIFT == 1 ? CLT_fft_bitrev240
             : SHIFT == 2 ? CLT_fft_bitrev120 : CLT_fft_bitrev60;
    t.window = CLT_window120;
    t.bitrev_sw = CLT_fft_bitrev480_sw;
    return t;
}

// LDS copy of one shift's tables (staged once per workgroup by the MDCT-only kernels).
struct MdctLds {
    u32 tw[480];
    u32 trig[480];
    i16 bitrev[480];
    i16 window[120];
};

template <int SHIFT>
CA_DEV MdctTab mdct_stage_tables(MdctLds &L, int tid, int nthreads)
{
    constexpr int N4 = 480 >> SHIFT;
    MdctTab g = mdct_global_tab<SHIFT>();
    for (int i = tid; i < 480; i += nthreads) L.tw[i] = g.tw[i];
    for (int i = tid; i < N4; i += nthreads) { L.trig[i] = g.trig[i]; L.bitrev[i] = g.bitrev[i]; }
    for (int i = tid; i < 120; i += nthreads) L.window[i] = g.window[i];
    MdctTab t;
    t.tw = L.tw; t.trig = L.trig; t.bitrev = L.bitrev; t.window = L.window;
    t.bitrev_sw = g.bitrev_sw;
    return t;
}

// ---- complex helpers ------------------------------------------------------------------------------
struct cpx { i32 r, i; };
CA_DEV cpx ld(const int2 *p) { int2 v = *p; return cpx{v.x, v.y}; }
CA_DEV void st(int2 *p, cpx v) { *p = make_int2(v.r, v.i); }
CA_DEV i32 lo16(u32 p) { return (i32)(i16)(p & 0xffff); }
CA_DEV i32 hi16(u32 p) { return (i32)p >> 16; }
#define CA_SMUL(x, t) mul16_32_q15((t), (x))       // S_MUL(a,b) = MULT16_32_Q15(b,a)

CA_DEV cpx c_mul(cpx a, u32 tw)                     // _kiss_fft_guts.h:59 C_MUL
{
    i32 tr = lo16(tw), ti = hi16(tw);
    cpx m;
    m.r = sub32(CA_SMUL(a.r, tr), CA_SMUL(a.i, ti));
    m.i = add32(CA_SMUL(a.r, ti), CA_SMUL(a.i, tr));
    return m;
}
CA_DEV cpx c_add(cpx a, cpx b) { return cpx{add32(a.r, b.r), add32(a.i, b.i)}; }
CA_DEV cpx c_sub(cpx a, cpx b) { return cpx{sub32(a.r, b.r), sub32(a.i, b.i)}; }

// The trips of a stage's lane loop stay rolled: unrolled, the loads of two or three trips are in flight together and the
// transform needs ~120 VGPRs (four wavefronts per SIMD); rolled it fits the budget of six to eight, and it is the other
// wavefronts of the SIMD, not the next trip of the same one, that cover a trip's LDS round trip.
#if defined(CA_HOST_EMU)
#define CA_FFT_ROLLED
CA_DEV int fft_lane(int lane) { return lane; }
#else
#define CA_FFT_ROLLED _Pragma("clang loop unroll(disable)")
// the lane index as a value the compiler cannot see through: every stage derives its addresses from it afresh, so the
// address arithmetic of a later stage is not hoisted above the earlier ones (where it would only hold registers)
CA_DEV int fft_lane(int lane) { asm volatile("" : "+v"(lane)); return lane; }
#endif

// ---- bank swizzle of the FFT scratch (long transforms) ---------------------------------------------------
// Point e of a 480-point transform lives at x[fsw<0>(e)]. Laid out plainly, the butterfly strides of the first stages
// (4, 8, 32 points x 8 bytes) and the bit-reversed scatter of the pre-rotation (fifteen consecutive inputs land 32 points
// apart) put most lanes of a wave-instruction on a few LDS banks: 570 conflict cycles on 277 useful ones per transform
// (measured ratio 2.6 on the frame kernel, profiles/r02_s7). The permutation
//     fsw(e) = e ^ (LUT[e >> 5] << 1) ^ (((e >> 4) & 1) << 2)
// keeps bit 0 (a pair of points stays one aligned 16-byte unit), permutes inside aligned groups of 32 points (so the
// lane-contiguous sweeps of the later stages stay conflict-free) and was searched (tools/fft_swizzle_search.py, the
// bank model of MI355X_MICROARCH.md: lane groups and banks per ds_read/ds_write width) for the access patterns below:
// pair scatter (b128), eight points per lane (b128), radix-4 at stride 8, radix-3 at 32, radix-5 at 96, linear read-out
// -- 12 conflict cycles left per transform, all in the scatter. Other sizes (short blocks, the hooks' 240 / 120) keep the plain layout.
template <int SHIFT>
CA_DEV int fsw(int e)
{
    if constexpr (SHIFT == 0) {
        // LUT[t], t = e >> 5 = 0..14, four bits each, pre-shifted left by one: CLT_FSW_LUT (celt_tables.h, tools/gen_tables.py)
        constexpr unsigned long long LUT = CLT_FSW_LUT;
        const int s = (int)(LUT >> ((e >> 3) & 60)) & 30;
        return e ^ s ^ ((e >> 2) & 4);
    } else {
        return e;
    }
}

// The P members e, e + M, ..., e + (P - 1) M of a butterfly. The transforms are issue-bound on the vector ALU (1 850 wave
// instructions each, PMC), so the swizzle is not evaluated member by member: the members of a butterfly share the 64-bit LUT
// shifted to the first one's block, differ in which 32-point block they fall in by a compile-time amount, and bit 4 of the index
// is the same for all of them (M = 32, 96) or a compile-time function of the member (M = 8).
template <int SW, int P, int M>
CA_DEV void fft_members(int e, int (&a)[P])
{
    if constexpr (SW != 0) {
#pragma unroll
        for (int c = 0; c < P; c++) a[c] = e + c * M;
    } else if constexpr (M == 8) {
        // e = 32 g + j, j < 8: one block, s the same for the four; e + 8 c = e ^ 8 c, and bit 4 of it is c >> 1
        const int A = e ^ ((int)(CLT_FSW_LUT >> ((e >> 3) & 60)) & 30);
#pragma unroll
        for (int c = 0; c < P; c++) a[c] = A ^ ((8 * c) ^ (4 * (c >> 1)));
    } else {
        static_assert(M % 32 == 0, "members whole blocks apart");
        const unsigned long long W = CLT_FSW_LUT >> ((e >> 3) & 60);
        const int E = e ^ ((e >> 2) & 4);
#pragma unroll
        for (int c = 0; c < P; c++) a[c] = (E + c * M) ^ ((int)(W >> (4 * ((c * M) >> 5))) & 30);
    }
}

// a pair of points (16 bytes, e even) in one LDS access
struct cpx2 { cpx a, b; };
CA_DEV cpx2 ld2(const int2 *p)
{
    const int4 v = *reinterpret_cast<const int4 *>(p);
    return cpx2{cpx{v.x, v.y}, cpx{v.z, v.w}};
}
CA_DEV void st2(int2 *p, cpx a, cpx b)
{
    int4 v;
    v.x = a.r; v.y = a.i; v.z = b.r; v.w = b.i;
    *reinterpret_cast<int4 *>(p) = v;
}

// ---- butterfly stages: B blocks of NFFT points laid out back to back in x[] ----------------------
template <int NFFT, int B>
CA_DEV void fft_radix4_first(int2 *x, int lane)     // kiss_fft.c:123-145, m == 1
{
    constexpr int CNT = B * NFFT / 4;
    for (int idx = lane; idx < CNT; idx += LANES) {
        int2 *f = x + 4 * idx;
        cpx x0 = ld(f), x1 = ld(f + 1), x2 = ld(f + 2), x3 = ld(f + 3);
        cpx s0 = c_sub(x0, x2);
        x0 = c_add(x0, x2);
        cpx s1 = c_add(x1, x3);
        x2 = c_sub(x0, s1);
        x0 = c_add(x0, s1);
        s1 = c_sub(x1, x3);
        st(f, x0);
        st(f + 2, x2);
        st(f + 1, cpx{add32(s0.r, s1.i), sub32(s0.i, s1.r)});
        st(f + 3, cpx{sub32(s0.r, s1.i), add32(s0.i, s1.r)});
    }
}

template <int NFFT, int B>
CA_DEV void fft_radix2_m4(int2 *x, int lane)        // kiss_fft.c:72-107, m == 4
{
    // one lane per (group, k) pair: k selects which of the four fixed rotations applies
    constexpr int CNT = B * NFFT / 2;
    constexpr i32 TW = 23170;
    for (int idx = lane; idx < CNT; idx += LANES) {
        int g = idx >> 2, k = idx & 3;
        int2 *f = x + 8 * g + k;
        cpx a = ld(f), b = ld(f + 4), t;
        if (k == 0) {
            t = b;
        } else if (k == 1) {
            t.r = CA_SMUL(add32(b.r, b.i), TW);
            t.i = CA_SMUL(sub32(b.i, b.r), TW);
        } else if (k == 2) {
            t.r = b.i;
            t.i = neg32(b.r);
        } else {
            t.r = CA_SMUL(sub32(b.i, b.r), TW);
            t.i = CA_SMUL(sub32(neg32(b.i), b.r), TW);
        }
        st(f + 4, c_sub(a, t));
        st(f, c_add(a, t));
    }
}

// The first two stages of the 480-point transform -- radix 4 at m == 1 (kiss_fft.c:123-145) and radix 2 at m == 4
// (:72-107) -- on eight consecutive points held by ONE lane: four 16-byte loads, twelve butterflies in registers, four
// 16-byte stores; 60 lanes, one LDS round trip instead of two (and no 4- and 8-point strides across lanes).
template <int SHIFT>
CA_DEV void fft_first8(int2 *x, int lane)
{
    constexpr int NFFT = 480 >> SHIFT;
    constexpr i32 TW = 23170;
    lane = fft_lane(lane);
    for (int g = lane; g < NFFT / 8; g += LANES) {
        cpx v[8];
        const int A = fsw<SHIFT>(8 * g);                                // the four pairs of a lane: fsw(8 g + 2 q) = fsw(8 g) ^ 2 q
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const cpx2 t = ld2(x + (A ^ (2 * q)));
            v[2 * q] = t.a;
            v[2 * q + 1] = t.b;
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {                                  // radix 4, m == 1, on points 4h .. 4h+3
            cpx x0 = v[4 * h], x1 = v[4 * h + 1], x2 = v[4 * h + 2], x3 = v[4 * h + 3];
            cpx s0 = c_sub(x0, x2);
            x0 = c_add(x0, x2);
            cpx s1 = c_add(x1, x3);
            x2 = c_sub(x0, s1);
            x0 = c_add(x0, s1);
            s1 = c_sub(x1, x3);
            v[4 * h] = x0;
            v[4 * h + 2] = x2;
            v[4 * h + 1] = cpx{add32(s0.r, s1.i), sub32(s0.i, s1.r)};
            v[4 * h + 3] = cpx{sub32(s0.r, s1.i), add32(s0.i, s1.r)};
        }
        {                                                               // radix 2, m == 4: (k, k + 4), rotation by k eighths
            cpx t, b;
            b = v[4]; t = b;
            v[4] = c_sub(v[0], t); v[0] = c_add(v[0], t);
            b = v[5]; t.r = CA_SMUL(add32(b.r, b.i), TW); t.i = CA_SMUL(sub32(b.i, b.r), TW);
            v[5] = c_sub(v[1], t); v[1] = c_add(v[1], t);
            b = v[6]; t.r = b.i; t.i = neg32(b.r);
            v[6] = c_sub(v[2], t); v[2] = c_add(v[2], t);
            b = v[7]; t.r = CA_SMUL(sub32(b.i, b.r), TW); t.i = CA_SMUL(sub32(neg32(b.i), b.r), TW);
            v[7] = c_sub(v[3], t); v[3] = c_add(v[3], t);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) st2(x + (A ^ (2 * q)), v[2 * q], v[2 * q + 1]);
    }
}

template <int NFFT, int B, int M>
CA_DEV void fft_radix4(int2 *x, const u32 *tw, int lane)   // kiss_fft.c:146-176
{
    constexpr int G = NFFT / (4 * M);
    constexpr int TWS = G * (480 / NFFT);
    constexpr int CNT = B * NFFT / 4;
    constexpr int SW = (NFFT == 480 && B == 1) ? 0 : 1;                 // layout: swizzled for the long transform only
    lane = fft_lane(lane);
    CA_FFT_ROLLED
    for (int idx = lane; idx < CNT; idx += LANES) {
        int blk = idx / (G * M), r = idx % (G * M);
        int g = r / M, j = r % M;
        const int e = blk * NFFT + g * 4 * M + j;
        int m[4];
        fft_members<SW, 4, M>(e, m);
        int2 *f0p = x + m[0], *f1p = x + m[1], *f2p = x + m[2], *f3p = x + m[3];
        cpx f0 = ld(f0p);
        cpx a = c_mul(ld(f1p), tw[j * TWS]);
        cpx b = c_mul(ld(f2p), tw[2 * j * TWS]);
        cpx c = c_mul(ld(f3p), tw[3 * j * TWS]);
        cpx d5 = c_sub(f0, b);
        f0 = c_add(f0, b);
        cpx s3 = c_add(a, c);
        cpx s4 = c_sub(a, c);
        st(f2p, c_sub(f0, s3));
        st(f0p, c_add(f0, s3));
        st(f1p, cpx{add32(d5.r, s4.i), sub32(d5.i, s4.r)});
        st(f3p, cpx{sub32(d5.r, s4.i), add32(d5.i, s4.r)});
    }
}

template <int NFFT, int B, int M>
CA_DEV void fft_radix3(int2 *x, const u32 *tw, int lane)   // kiss_fft.c:185-241
{
    constexpr int G = NFFT / (3 * M);
    constexpr int TWS = G * (480 / NFFT);
    constexpr int CNT = B * NFFT / 3;
    constexpr int SW = (NFFT == 480 && B == 1) ? 0 : 1;
    constexpr i32 EPI3 = -28378;
    lane = fft_lane(lane);
    CA_FFT_ROLLED
    for (int idx = lane; idx < CNT; idx += LANES) {
        int blk = idx / (G * M), r = idx % (G * M);
        int g = r / M, j = r % M;
        const int e = blk * NFFT + g * 3 * M + j;
        int m[3];
        fft_members<SW, 3, M>(e, m);
        int2 *f0p = x + m[0], *f1p = x + m[1], *f2p = x + m[2];
        cpx f0 = ld(f0p);
        cpx a = c_mul(ld(f1p), tw[j * TWS]);
        cpx b = c_mul(ld(f2p), tw[2 * j * TWS]);
        cpx s3 = c_add(a, b);
        cpx s0 = c_sub(a, b);
        cpx f1{sub32(f0.r, s3.r >> 1), sub32(f0.i, s3.i >> 1)};
        s0.r = CA_SMUL(s0.r, EPI3);
        s0.i = CA_SMUL(s0.i, EPI3);
        st(f0p, c_add(f0, s3));
        st(f2p, cpx{add32(f1.r, s0.i), sub32(f1.i, s0.r)});
        st(f1p, cpx{sub32(f1.r, s0.i), add32(f1.i, s0.r)});
    }
}

template <int NFFT, int B, int M>
CA_DEV void fft_radix5(int2 *x, const u32 *tw, int lane)   // kiss_fft.c:245-322
{
    constexpr int G = NFFT / (5 * M);
    constexpr int TWS = G * (480 / NFFT);
    constexpr int CNT = B * NFFT / 5;
    constexpr int SW = (NFFT == 480 && B == 1) ? 0 : 1;
    constexpr i32 YAR = 10126, YAI = -31164, YBR = -26510, YBI = -19261;
    lane = fft_lane(lane);
    CA_FFT_ROLLED
    for (int idx = lane; idx < CNT; idx += LANES) {
        int blk = idx / (G * M), r = idx % (G * M);
        int g = r / M, u = r % M;
        const int e = blk * NFFT + g * 5 * M + u;
        int m[5];
        fft_members<SW, 5, M>(e, m);
        int2 *f0p = x + m[0], *f1p = x + m[1], *f2p = x + m[2], *f3p = x + m[3], *f4p = x + m[4];
        cpx s0 = ld(f0p);
        cpx s1 = c_mul(ld(f1p), tw[u * TWS]);
        cpx s2 = c_mul(ld(f2p), tw[2 * u * TWS]);
        cpx s3 = c_mul(ld(f3p), tw[3 * u * TWS]);
        cpx s4 = c_mul(ld(f4p), tw[4 * u * TWS]);
        cpx s7 = c_add(s1, s4), s10 = c_sub(s1, s4);
        cpx s8 = c_add(s2, s3), s9 = c_sub(s2, s3);
        st(f0p, cpx{add32(s0.r, add32(s7.r, s8.r)), add32(s0.i, add32(s7.i, s8.i))});
        cpx s5, s6, s11, s12;
        s5.r = add32(add32(s0.r, CA_SMUL(s7.r, YAR)), CA_SMUL(s8.r, YBR));
        s5.i = add32(add32(s0.i, CA_SMUL(s7.i, YAR)), CA_SMUL(s8.i, YBR));
        s6.r = add32(CA_SMUL(s10.i, YAI), CA_SMUL(s9.i, YBI));
        s6.i = sub32(neg32(CA_SMUL(s10.r, YAI)), CA_SMUL(s9.r, YBI));
        st(f1p, c_sub(s5, s6));
        st(f4p, c_add(s5, s6));
        s11.r = add32(add32(s0.r, CA_SMUL(s7.r, YBR)), CA_SMUL(s8.r, YAR));
        s11.i = add32(add32(s0.i, CA_SMUL(s7.i, YBR)), CA_SMUL(s8.i, YAR));
        s12.r = add32(neg32(CA_SMUL(s10.i, YBI)), CA_SMUL(s9.i, YAI));
        s12.i = sub32(CA_SMUL(s10.r, YBI), CA_SMUL(s9.r, YAI));
        st(f2p, c_add(s11, s12));
        st(f3p, c_sub(s11, s12));
    }
}

// Where point e of the scratch lives: the long single transform is bank-swizzled, everything else plain.
template <int SHIFT, int B> CA_DEV int fft_at(int e) { return fsw<(SHIFT == 0 && B == 1) ? 0 : 1>(e); }
template <int SHIFT, int B> CA_DEV void fft_put(int2 *x, int e, cpx v) { st(x + fft_at<SHIFT, B>(e), v); }
template <int SHIFT, int B> CA_DEV cpx fft_get(const int2 *x, int e) { return ld(x + fft_at<SHIFT, B>(e)); }

// In-place FFT of B blocks of (480 >> SHIFT) bit-reversed points (opus_fft_impl, kiss_fft.c:532-578); the points are
// addressed through fft_put / fft_get.
template <int SHIFT, int B>
CA_DEV void fft_wave(int2 *x, const u32 *tw, int lane)
{
    constexpr int NFFT = 480 >> SHIFT;
    if constexpr (SHIFT == 0 && B == 1) {
        // swizzled layout (fsw<0>): points written by fft_put<0, 1> / the MDCT pre-rotations, read back with fft_get<0, 1>
        fft_first8<0>(x, lane);                 wave_sync();
        fft_radix4<NFFT, B, 8>(x, tw, lane);    wave_sync();
        fft_radix3<NFFT, B, 32>(x, tw, lane);   wave_sync();
        fft_radix5<NFFT, B, 96>(x, tw, lane);   wave_sync();
        return;
    }
    fft_radix4_first<NFFT, B>(x, lane);
    wave_sync();
    if constexpr (SHIFT == 0) {
        fft_radix2_m4<NFFT, B>(x, lane);        wave_sync();
        fft_radix4<NFFT, B, 8>(x, tw, lane);    wave_sync();
        fft_radix3<NFFT, B, 32>(x, tw, lane);   wave_sync();
        fft_radix5<NFFT, B, 96>(x, tw, lane);   wave_sync();
    } else if constexpr (SHIFT == 1) {
        fft_radix4<NFFT, B, 4>(x, tw, lane);    wave_sync();
        fft_radix3<NFFT, B, 16>(x, tw, lane);   wave_sync();
        fft_radix5<NFFT, B, 48>(x, tw, lane);   wave_sync();
    } else if constexpr (SHIFT == 2) {
        fft_radix2_m4<NFFT, B>(x, lane);        wave_sync();
        fft_radix3<NFFT, B, 8>(x, tw, lane);    wave_sync();
        fft_radix5<NFFT, B, 24>(x, tw, lane);   wave_sync();
    } else {
        fft_radix3<NFFT, B, 4>(x, tw, lane);    wave_sync();
        fft_radix5<NFFT, B, 12>(x, tw, lane);   wave_sync();
    }
}

// ---- MDCT ------------------------------------------------------------------------------------------
// Forward MDCT of B blocks (hop N2, block b reads sin[b*N2 .. b*N2+N2+120)); coefficient k of block b
// goes to dst[(k*B + b) * dstride] -- the interleaved layout of compute_mdcts (celt_encoder.c:441).
// T holds the tables for this SHIFT. dst may be LDS or global; it may alias sin
// (all reads of sin complete before the first write to dst).      Reference: mdct.c:121-259.
// fold + pre-rotation of input point i of one block (mdct.c:155-231): the value the FFT takes at position bitrev[i]
template <int SHIFT>
CA_DEV cpx mdct_fwd_pre(const i32 *in, int i, const MdctTab &T)
{
    constexpr int N2 = 960 >> SHIFT, N4 = N2 / 2;
    constexpr int OV = 120, OV2 = 60, Q = 30;
    constexpr int SCALE_SHIFT = (8 - SHIFT) - 1;                     // st->scale_shift - 1
    const int a = OV2 + 2 * i, b = N2 - 1 + OV2 - 2 * i;
    i32 re, im;
    if (i < Q) {                                                      // mdct.c:162-175
        i32 w1 = T.window[OV2 + 2 * i], w2 = T.window[OV2 - 1 - 2 * i];
        re = add32(mul16_32_q15(w2, in[a + N2]), mul16_32_q15(w1, in[b]));
        im = sub32(mul16_32_q15(w1, in[a]), mul16_32_q15(w2, in[b - N2]));
    } else if (i < N4 - Q) {                                          // mdct.c:178-189
        re = in[b];
        im = in[a];
    } else {                                                          // mdct.c:190-203
        int k = i - (N4 - Q);
        i32 w1 = T.window[2 * k], w2 = T.window[OV - 1 - 2 * k];
        re = add32(neg32(mul16_32_q15(w1, in[a - N2])), mul16_32_q15(w2, in[b]));
        im = add32(mul16_32_q15(w2, in[a]), mul16_32_q15(w1, in[b + N2]));
    }
    u32 t = T.trig[i];                                                // mdct.c:206-231
    i32 t0 = lo16(t), t1 = hi16(t);
    i32 yr = sub32(CA_SMUL(re, t0), CA_SMUL(im, t1));
    i32 yi = add32(CA_SMUL(im, t0), CA_SMUL(re, t1));
    return cpx{pshr32(mul16_32_q16(17476, yr), SCALE_SHIFT), pshr32(mul16_32_q16(17476, yi), SCALE_SHIFT)};
}

// Lane schedule of the pre-rotation of the long transform: inputs i and i + 120 land on neighbouring points (bitrev[i + 120]
// == bitrev[i] + 1 for i in [0, 120) and [240, 360)), so one lane produces both and stores them as ONE 16-byte unit; 240
// such pairs, 60 per trip (the lanes of a trip read consecutive inputs: same coalescing as a plain sweep).
enum { FFT_PAIRS = 240, FFT_PAIR_STEP = LANES >= 60 ? 60 : LANES };
CA_DEV int fft_pair_input(int n) { return n < 120 ? n : n + 120; }

template <int SHIFT, int B>
CA_DEV void mdct_forward_wave(const i32 *sin, int2 *f2, i32 *dst, int dstride, const MdctTab &T, int lane)
{
    const u32 *trig = T.trig;
    const i16 *bitrev = T.bitrev;
    constexpr int N2 = 960 >> SHIFT, N4 = N2 / 2, NFFT = N4;
    if constexpr (SHIFT == 0 && B == 1) {
        static_assert(FFT_PAIRS % FFT_PAIR_STEP == 0, "whole trips");
        if (lane < FFT_PAIR_STEP) {
            CA_FFT_ROLLED
            for (int n0 = 0; n0 < FFT_PAIRS; n0 += FFT_PAIR_STEP) {
                const int i = fft_pair_input(n0 + lane);
                const cpx a = mdct_fwd_pre<SHIFT>(sin, i, T), b = mdct_fwd_pre<SHIFT>(sin, i + 120, T);
                st2(f2 + T.bitrev_sw[i], a, b);
            }
        }
    } else {
        for (int idx = lane; idx < B * N4; idx += LANES) {
            int blk = idx / N4, i = idx % N4;
            fft_put<SHIFT, B>(f2, blk * NFFT + bitrev[i], mdct_fwd_pre<SHIFT>(sin + blk * N2, i, T));
        }
    }
    wave_sync();
    fft_wave<SHIFT, B>(f2, T.tw, lane);
    for (int idx = lane; idx < B * N4; idx += LANES) {                  // mdct.c:237-257
        int blk = idx / N4, i = idx % N4;
        cpx f = fft_get<SHIFT, B>(f2, blk * NFFT + i);
        u32 t = trig[i];
        i32 t0 = lo16(t), t1 = hi16(t);
        dst[((2 * i) * B + blk) * dstride] = sub32(CA_SMUL(f.i, t1), CA_SMUL(f.r, t0));
        dst[((N2 - 1 - 2 * i) * B + blk) * dstride] = add32(CA_SMUL(f.r, t1), CA_SMUL(f.i, t0));
    }
    wave_sync();
}

// Inverse MDCT of B blocks. Coefficient k of block b is src[(k*B + b) * sstride]; block b produces
// out[b*N2 + 60 .. b*N2 + 60 + N2) and TDAC-mixes out[b*N2 .. b*N2+120) (the first 60 samples of
// block 0 are the previous frame's tail). src must not alias out.    Reference: mdct.c:263-363.
template <int SHIFT, int B>
CA_DEV void mdct_backward_wave(const i32 *src, int sstride, int2 *f2, i32 *out, const MdctTab &T, int lane)
{
    const u32 *trig = T.trig;
    const i16 *bitrev = T.bitrev;
    constexpr int N2 = 960 >> SHIFT, N4 = N2 / 2, NFFT = N4;
    constexpr int OV = 120, OV2 = 60;
    auto pre = [&](int blk, int i) {                                    // mdct.c:283-304
        i32 x1 = src[((2 * i) * B + blk) * sstride];
        i32 x2 = src[((N2 - 1 - 2 * i) * B + blk) * sstride];
        u32 t = trig[i];
        i32 t0 = lo16(t), t1 = hi16(t);
        i32 yr = add32(CA_SMUL(x2, t0), CA_SMUL(x1, t1));
        i32 yi = sub32(CA_SMUL(x1, t0), CA_SMUL(x2, t1));
        return cpx{yi, yr};                                             // re/im swapped: FFT as IFFT
    };
    if constexpr (SHIFT == 0 && B == 1) {
        if (lane < FFT_PAIR_STEP) {                                    // pairs (i, i + 120): see mdct_forward_wave
            CA_FFT_ROLLED
            for (int n0 = 0; n0 < FFT_PAIRS; n0 += FFT_PAIR_STEP) {
                const int i = fft_pair_input(n0 + lane);
                st2(f2 + T.bitrev_sw[i], pre(0, i), pre(0, i + 120));
            }
        }
    } else {
        for (int idx = lane; idx < B * N4; idx += LANES) {
            int blk = idx / N4, i = idx % N4;
            fft_put<SHIFT, B>(f2, blk * NFFT + bitrev[i], pre(blk, i));
        }
    }
    wave_sync();
    fft_wave<SHIFT, B>(f2, T.tw, lane);
    // post-rotate + de-shuffle (mdct.c:310-342). The reference walks the buffer from both ends in
    // place; element k always lands as y[2k] = yr(k), y[2(N4-1-k)+1] = yi(k) with twiddles
    // trig[k], trig[N4+k], which is what each lane computes here.
    for (int idx = lane; idx < B * N4; idx += LANES) {
        int blk = idx / N4, k = idx % N4;
        cpx f = fft_get<SHIFT, B>(f2, blk * NFFT + k);
        i32 re = f.i, im = f.r;
        u32 t = trig[k];
        i32 t0 = lo16(t), t1 = hi16(t);
        i32 *y = out + blk * N2 + OV2;
        y[2 * k] = add32(CA_SMUL(re, t0), CA_SMUL(im, t1));
        y[2 * (N4 - 1 - k) + 1] = sub32(CA_SMUL(re, t1), CA_SMUL(im, t0));
    }
    wave_sync();
    for (int idx = lane; idx < B * OV2; idx += LANES) {                 // mdct.c:345-361 TDAC mirror
        int blk = idx / OV2, i = idx % OV2;
        i32 *o = out + blk * N2;
        i32 x1 = o[OV - 1 - i], x2 = o[i];
        i32 w1 = T.window[i], w2 = T.window[OV - 1 - i];
        o[i] = sub32(mul16_32_q15(w2, x2), mul16_32_q15(w1, x1));
        o[OV - 1 - i] = add32(mul16_32_q15(w1, x2), mul16_32_q15(w2, x1));
    }
    wave_sync();
}

}  // namespace ca

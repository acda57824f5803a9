// celt_stage_kernels.hip -- lane-per-(frame, channel) kernels for the strictly serial stages of the front
// phase (see celt_stage_lane.h): 64 independent recurrences per wavefront instead of 2.
#include "celt_enc.h"
#include "celt_stage_lane.h"
#include "opusgpu_internal.h"
#include <stdlib.h>

namespace ca {

static_assert(offsetof(FrameMid, X) % 16 == 0 && sizeof(FrameMid) % 16 == 0, "FrameMid::X is moved 16 bytes at a time");

// dc_reject: pcm [n][960][2] int16 -> mid[f].X planar [2][960] int16, filter memory -> mid[f].hp_mem
__global__ __launch_bounds__(256) void celt_dc_reject_kernel(const opusgpu_celt_state *states, const i16 *__restrict__ pcm,
                                                             FrameMid *__restrict__ mid, int nframes)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int f = t >> 1, c = t & 1;
    if (f >= nframes) return;
    i32 hp[2] = {0, 0};
    if (states) { hp[0] = states[f].hp_mem[2 * c]; hp[1] = states[f].hp_mem[2 * c + 1]; }
    stage_dc_reject_channel(pcm + (size_t)f * FRAME * 2, c, hp, mid[f].X + c * FRAME);
    mid[f].hp_mem[2 * c] = hp[0];
    mid[f].hp_mem[2 * c + 1] = hp[1];
}

// transient metric per channel: in_ws [n][2][1080] int32 -> mid[f].trans_unmask[c]; mid[f].X is scratch here
__global__ __launch_bounds__(256) void celt_transient_kernel(FrameMid *__restrict__ mid, const i32 *__restrict__ in_ws, int nframes)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int f = t >> 1, c = t & 1;
    if (f >= nframes) return;
    mid[f].trans_unmask[c] = stage_transient_channel(in_ws + ((size_t)f * 2 + c) * (FRAME + OVL), mid[f].X + c * FRAME);
}

// ---- transient metric, tiled through LDS ---------------------------------------------------------------------------
// Same arithmetic as stage_transient_channel (celt_stage_lane.h), different data movement. One wavefront = 64 rows
// (32 frames x 2 channels) of in_ws. Per lane the row is a 4 320-byte stream at a 4 320-byte stride from its
// neighbours': read 16 bytes at a time every lane pulls its own cache line through L1 again and again (measured 2.6 GB
// of HBM traffic per 65 536 frames for 0.57 GB of input). Here the wavefront loads the rows tile by tile (40 samples x
// 64 rows, ten consecutive lanes per 160-byte row segment, the next tile in flight while the current one is
// consumed), each lane then walks its row of the tile in LDS, and the follower array (540 int16 per row) never leaves
// LDS. 11 KB (tile) + 67.5 KB (followers) per wavefront -> two wavefronts per CU, which is enough: the kernel is bound
// by the two recurrences, 2 x 1 080 dependent steps per row.
enum { TT = 40, TW = TT + 4, NTILE = (FRAME + OVL) / TT, TROWS = 64, TLEN2 = (FRAME + OVL) / 2 };
static_assert(NTILE * TT == FRAME + OVL && TT % 8 == 0 && (TW / 4) % 2 == 1, "tile shape");

struct __attribute__((aligned(16))) TransLds {
    i32 tile[TROWS][TW];
    i16 fol[TROWS][TLEN2];
};

struct TileRegs { int4 v[TT * TROWS / 4 / 64]; };      // 10 chunks of 16 bytes per lane

CA_DEV void trans_tile_fetch(TileRegs &R, const i32 *__restrict__ rows0, int t, int nrows, int lane)
{
#pragma unroll
    for (int m = 0; m < TT * TROWS / 4 / 64; m++) {
        const int q = lane + 64 * m, row = q / (TT / 4), col = q % (TT / 4);
        R.v[m] = make_int4(0, 0, 0, 0);
        if (row < nrows) R.v[m] = *reinterpret_cast<const int4 *>(rows0 + (size_t)row * (FRAME + OVL) + t * TT + col * 4);
    }
}

CA_DEV void trans_tile_store(TransLds &S, const TileRegs &R, int lane)
{
#pragma unroll
    for (int m = 0; m < TT * TROWS / 4 / 64; m++) {
        const int q = lane + 64 * m, row = q / (TT / 4), col = q % (TT / 4);
        *reinterpret_cast<int4 *>(&S.tile[row][col * 4]) = R.v[m];
    }
}

__global__ __launch_bounds__(64) void celt_transient_tile_kernel(FrameMid *__restrict__ mid, const i32 *__restrict__ in_ws, int nframes)
{
    __shared__ TransLds S;
    const int lane = threadIdx.x;
    const int row0 = blockIdx.x * TROWS, total = 2 * nframes;
    const int nrows = total - row0 < TROWS ? total - row0 : TROWS;
    const i32 *rows0 = in_ws + (size_t)row0 * (FRAME + OVL);
    const bool live = lane < nrows;
    const int len = FRAME + OVL, len2 = TLEN2;
    TileRegs R;
    // pass 1: high-pass, extrema (celt_encoder.c:262-283)
    i32 mem0 = 0, mem1 = 0, mx = 0, mn = 0;
    trans_tile_fetch(R, rows0, 0, nrows, lane);
    for (int t = 0; t < NTILE; t++) {
        wave_sync();
        trans_tile_store(S, R, lane);
        wave_sync();
        if (t + 1 < NTILE) trans_tile_fetch(R, rows0, t + 1, nrows, lane);
#pragma unroll 2
        for (int k0 = 0; k0 < TT; k0 += 4) {
            const int4 a = *reinterpret_cast<const int4 *>(&S.tile[lane][k0]);
            const i32 w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                i32 v = stage_trans_hp(w[k], mem0, mem1);
                if (t * TT + k0 + k < 12) v = 0;
                mx = imax(mx, v);
                mn = imin(mn, v);
            }
        }
    }
    const int shift = 14 - celt_ilog2(1 + imax(mx, -mn));
    // pass 2: high-pass again, normalise, pair energies, forward follower -> fol (celt_encoder.c:285-308)
    i32 mean = 0, fm = 0;
    mem0 = mem1 = 0;
    trans_tile_fetch(R, rows0, 0, nrows, lane);
    for (int t = 0; t < NTILE; t++) {
        wave_sync();
        trans_tile_store(S, R, lane);
        wave_sync();
        if (t + 1 < NTILE) trans_tile_fetch(R, rows0, t + 1, nrows, lane);
#pragma unroll 1
        for (int k0 = 0; k0 < TT; k0 += 8) {
            const int4 a = *reinterpret_cast<const int4 *>(&S.tile[lane][k0]), b = *reinterpret_cast<const int4 *>(&S.tile[lane][k0 + 4]);
            const i32 w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            i32 tv[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                tv[k] = stage_trans_hp(w[k], mem0, mem1);
                if (t * TT + k0 + k < 12) tv[k] = 0;
                if (shift != 0) tv[k] = (i16)shl16(tv[k], shift);
            }
            u32 f[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                i32 x2 = (i16)pshr32(add32(mul16_16(tv[2 * k], tv[2 * k]), mul16_16(tv[2 * k + 1], tv[2 * k + 1])), 16);
                mean = add32(mean, x2);
                fm = (i16)(fm + pshr32(x2 - fm, 4));
                f[k] = (u32)fm & 0xffffu;
            }
            int2 r;
            r.x = (i32)(f[0] | (f[1] << 16));
            r.y = (i32)(f[2] | (f[3] << 16));
            *reinterpret_cast<int2 *>(&S.fol[lane][(t * TT + k0) >> 1]) = r;
        }
    }
    wave_sync();
    // pass 3: backward follower in place (celt_encoder.c:310-322); len2 = 540 = 135 groups of 4
    i32 bm = 0, maxE = 0;
#pragma unroll 5
    for (int g = len2 / 4 - 1; g >= 0; g--) {
        int2 v = *reinterpret_cast<const int2 *>(&S.fol[lane][4 * g]);
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
        *reinterpret_cast<int2 *>(&S.fol[lane][4 * g]) = r;
    }
    // pass 4: masking metric (celt_encoder.c:324-352)
    mean = mul16_16(celt_sqrt(mean), celt_sqrt(mul16_16(maxE, len2 >> 1)));
    const i32 norm = shl32(len2, 6 + 14) / add32(1, mean >> 1);
    i32 unmask = 0;
    for (int i = 12; i < len2 - 5; i += 4) {
        i32 id = imax(0, imin(127, mul16_32_q15((i16)(S.fol[lane][i] + 1), norm)));
        unmask += CLT_inv_table[id];
    }
    (void)len;
    if (live) {
        const int R0 = row0 + lane;
        mid[R0 >> 1].trans_unmask[R0 & 1] = 64 * unmask * 4 / (6 * (len2 - 17));
    }
}

}  // namespace ca

extern "C" void opusgpu_launch_dc_reject(const void *states, const int16_t *pcm, void *mid, int n, hipStream_t s)
{
    hipLaunchKernelGGL(ca::celt_dc_reject_kernel, dim3((2 * n + 255) / 256), dim3(256), 0, s, (const opusgpu_celt_state *)states, pcm,
                       (ca::FrameMid *)mid, n);
}

extern "C" void opusgpu_launch_transient(void *mid, const int32_t *in_ws, int n, hipStream_t s)
{
    // default: rows tiled through LDS; OPUSGPU_TRANSIENT_LANE=1 selects the streaming lane kernel (cross-check)
    if (getenv("OPUSGPU_TRANSIENT_LANE"))
        hipLaunchKernelGGL(ca::celt_transient_kernel, dim3((2 * n + 255) / 256), dim3(256), 0, s, (ca::FrameMid *)mid, in_ws, n);
    else
        hipLaunchKernelGGL(ca::celt_transient_tile_kernel, dim3((2 * n + 63) / 64), dim3(64), 0, s, (ca::FrameMid *)mid, in_ws, n);
}

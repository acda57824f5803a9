// silk_shape_kernels.hip -- batched silk_noise_shape_analysis_FIX (opus-fix/silk/fixed/noise_shape_analysis_FIX.c:146-466), one lane
// per frame. The arithmetic lives in silk_shape_dev.h; the windowed analysis block of each of the wavefront's 64 frames lives in
// LDS laid out [sample][lane].
#include <string.h>
#include "silk_shape_dev.h"
#include "silk_prefilter_dev.h"
#include "opusgpu_internal.h"
#include "../../include/opusgpu_silk.h"
#include "silk_validate.h"

namespace ca {

struct ShapeCol {                                              // this lane's column of a [sample][lane] block
    i16 *p;
    __device__ __forceinline__ i16 &operator[](int k) const { return p[k * 64]; }
    __device__ __forceinline__ ShapeCol operator+(int o) const { ShapeCol r; r.p = p + o * 64; return r; }
};

__global__ __launch_bounds__(64) void silk_noise_shape_kernel(const opusgpu_noise_shape_in *__restrict__ recs,
                                                              opusgpu_noise_shape_out *__restrict__ outs, int n_rec,
                                                              int *__restrict__ bad_records)
{
    // one block: the plain autocorrelation's down-shifted copy is made in place (the windowed block is not read again), which keeps
    // the workgroup at 30 KB of LDS -- four of them per CU, i.e. all 65 536 frames of a full batch resident at once
    __shared__ i16 xw_s[240 * 64];
    const int r = blockIdx.x * 64 + threadIdx.x;
    if (r >= n_rec) return;
    const opusgpu_noise_shape_in &in = recs[r];
    opusgpu_noise_shape_out &out = outs[r];
    if (!noise_shape_record_ok(in)) {
        memset(&out, 0, sizeof(out));
        out.status = OPUSGPU_BAD_ARG;
        atomicAdd(bad_records, 1);
        return;
    }
    ShapeCfg c;
    c.fs_kHz = in.fs_kHz; c.nb_subfr = in.nb_subfr; c.subfr_length = in.subfr_length; c.la_shape = in.la_shape; c.shapeWinLength = in.shapeWinLength;
    c.shapingLPCOrder = in.shapingLPCOrder; c.warping_Q16 = in.warping_Q16; c.SNR_dB_Q7 = in.SNR_dB_Q7; c.useCBR = in.useCBR;
    c.speech_activity_Q8 = in.speech_activity_Q8; c.signalType = in.signalType; c.input_quality_bands_Q15[0] = in.input_quality_bands_Q15[0];
    c.input_quality_bands_Q15[1] = in.input_quality_bands_Q15[1]; c.LTPCorr_Q15 = in.LTPCorr_Q15; c.predGain_Q16 = in.predGain_Q16;
    for (int k = 0; k < 4; k++) c.pitchL[k] = in.pitchL[k];
    ShapeOut o;
    memset(&o, 0, sizeof(o));
    o.HarmBoost_smth_Q16 = in.HarmBoost_smth_Q16; o.HarmShapeGain_smth_Q16 = in.HarmShapeGain_smth_Q16; o.Tilt_smth_Q16 = in.Tilt_smth_Q16;
    ShapeCol xw, xs;
    xw.p = xw_s + threadIdx.x;
    xs.p = xw_s + threadIdx.x;
    silk_noise_shape_analysis_dev(c, (const i16 *)in.pitch_res, (const i16 *)in.x + in.la_shape, xw, xs, o);
    memset(&out, 0, sizeof(out));
    for (int k = 0; k < in.nb_subfr; k++) { out.Gains_Q16[k] = o.Gains_Q16[k]; out.GainsPre_Q14[k] = o.GainsPre_Q14[k]; out.LF_shp_Q14[k] = o.LF_shp_Q14[k]; }
    for (int k = 0; k < 4 * MAX_SHAPE_LPC_ORDER; k++) { out.AR1_Q13[k] = o.AR1_Q13[k]; out.AR2_Q13[k] = o.AR2_Q13[k]; }
    for (int k = 0; k < 4; k++) { out.HarmBoost_Q14[k] = o.HarmBoost_Q14[k]; out.HarmShapeGain_Q14[k] = o.HarmShapeGain_Q14[k]; out.Tilt_Q14[k] = o.Tilt_Q14[k]; }
    out.HarmBoost_smth_Q16 = o.HarmBoost_smth_Q16; out.HarmShapeGain_smth_Q16 = o.HarmShapeGain_smth_Q16; out.Tilt_smth_Q16 = o.Tilt_smth_Q16;
    out.input_quality_Q14 = o.input_quality_Q14; out.coding_quality_Q14 = o.coding_quality_Q14; out.sparseness_Q8 = o.sparseness_Q8;
    out.quantOffsetType = o.quantOffsetType;
    out.status = OPUSGPU_OK;
}

// The harmonic-shaping ring of silk_prefilter_state_FIX (sLTP_shp[512]) as the kernel holds it: a frame reads at most
// lag + 2 <= PF_RING - 1 samples back from its newest entry and appends one per sample, so a lane keeps the newest PF_RING
// entries as a ring of its own in its LDS column (40 KB per workgroup instead of 64: four workgroups per CU, the whole batch in
// one round) and translates the 512-ring positions silk_prefilter_dev computes: distance from the newest entry -> slot.
enum { PF_RING = 320 };
struct PrefiltRingPos { int cur, head; };                      // 512-ring position and slot of the newest entry
struct PrefiltRing {
    i16 *col;                                                  // this lane's column, slot stride 64
    PrefiltRingPos *s;
    struct Ref {
        const PrefiltRing *r;
        int p;
        __device__ __forceinline__ operator i32() const
        {
            int t = r->s->head + ((p - r->s->cur) & LTP_MASK);
            t = t >= PF_RING ? t - PF_RING : t;
            return (i32)r->col[t * 64];
        }
        __device__ __forceinline__ void operator=(i16 v) const                              // the reference appends at (newest - 1) & 511
        {
            const int h = r->s->head == 0 ? PF_RING - 1 : r->s->head - 1;
            r->s->head = h;
            r->s->cur = p;
            r->col[h * 64] = v;
        }
    };
    __device__ __forceinline__ Ref operator[](int k) const { return Ref{this, k}; }
};

struct OutCol32 {                                              // xw_Q3 straight into the output record
    i32 *p;
    __device__ __forceinline__ i32 &operator[](int k) const { return p[k]; }
};

__global__ __launch_bounds__(64) void silk_prefilter_kernel(const opusgpu_prefilter_in *__restrict__ recs, opusgpu_prefilter_state *__restrict__ states,
                                                            opusgpu_prefilter_out *__restrict__ outs, int n_rec, int *__restrict__ bad_records)
{
    __shared__ i16 ltp_s[PF_RING * 64];                        // the newest PF_RING entries of the 64 harmonic-shaping rings, [slot][lane]
    const int r = blockIdx.x * 64 + threadIdx.x;
    if (r >= n_rec) return;
    const opusgpu_prefilter_in &in = recs[r];
    opusgpu_prefilter_state &st = states[r];
    opusgpu_prefilter_out &out = outs[r];
    if (!prefilter_record_ok(in, st)) {
        for (int k = 0; k < OPUSGPU_SILK_MAX_FRAME; k++) out.xw_Q3[k] = 0;
        out.status = OPUSGPU_BAD_ARG;
        atomicAdd(bad_records, 1);
        return;
    }
    PrefiltRingPos rp;
    rp.cur = st.sLTP_shp_buf_idx;
    rp.head = 0;
    PrefiltRing ltp;
    ltp.col = ltp_s + threadIdx.x;
    ltp.s = &rp;
    static_assert(PF_RING % 32 == 0, "whole batches");
    for (int a = 0; a < PF_RING; a += 32) {                    // slot a <- the entry a behind the newest, eight per access where the 512-ring does not
        i32 v[4][8];                                           // wrap; four accesses in flight (a load per trip would be an exposed round trip per trip)
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int p0 = (rp.cur + a + 8 * c) & LTP_MASK;
            if (p0 <= LTP_BUF_LENGTH - 8) pe_load8(v[c], (const i16 *)st.sLTP_shp + p0);
            else
                for (int u = 0; u < 8; u++) v[c][u] = st.sLTP_shp[(p0 + u) & LTP_MASK];
        }
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int u = 0; u < 8; u++) ltp.col[(a + 8 * c + u) * 64] = (i16)v[c][u];
    }
    PrefilterState P;
    for (int k = 0; k <= MAX_SHAPE_LPC_ORDER; k++) P.sAR_shp[k] = st.sAR_shp[k];
    P.sLTP_shp_buf_idx = st.sLTP_shp_buf_idx; P.sLF_AR_shp_Q12 = st.sLF_AR_shp_Q12; P.sLF_MA_shp_Q12 = st.sLF_MA_shp_Q12;
    P.sHarmHP_Q2 = st.sHarmHP_Q2; P.rand_seed = st.rand_seed; P.lagPrev = st.lagPrev;
    PrefilterCtrl c;
    for (int k = 0; k < 4; k++) {
        c.pitchL[k] = in.pitchL[k]; c.HarmShapeGain_Q14[k] = in.HarmShapeGain_Q14[k]; c.HarmBoost_Q14[k] = in.HarmBoost_Q14[k];
        c.Tilt_Q14[k] = in.Tilt_Q14[k]; c.GainsPre_Q14[k] = in.GainsPre_Q14[k]; c.LF_shp_Q14[k] = in.LF_shp_Q14[k];
    }
    for (int k = 0; k < 4 * MAX_SHAPE_LPC_ORDER; k++) c.AR1_Q13[k] = in.AR1_Q13[k];
    c.coding_quality_Q14 = in.coding_quality_Q14; c.nb_subfr = in.nb_subfr; c.subfr_length = in.subfr_length; c.signalType = in.signalType;
    c.warping_Q16 = in.warping_Q16; c.shapingLPCOrder = in.shapingLPCOrder;
    OutCol32 xw;
    xw.p = out.xw_Q3;
    silk_prefilter_dev(P, c, (const i16 *)in.x, xw, ltp);
    for (int k = in.nb_subfr * in.subfr_length; k < OPUSGPU_SILK_MAX_FRAME; k++) out.xw_Q3[k] = 0;
    {                                                          // the frame's new entries back into the 512-ring (the others have not changed)
        const int N = in.nb_subfr * in.subfr_length;           // <= PF_RING: appended this call, newest at rp.cur / slot rp.head
        for (int a = 0; a < N; a += 8) {
            const int p0 = (rp.cur + a) & LTP_MASK;
            i32 v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                int t = rp.head + a + u;
                t = t >= PF_RING ? t - PF_RING : t;
                v[u] = ltp.col[t * 64];
            }
            if (p0 <= LTP_BUF_LENGTH - 8 && a + 8 <= N) pe_store8((i16 *)st.sLTP_shp + p0, v);
            else
                for (int u = 0; u < 8 && a + u < N; u++) st.sLTP_shp[(p0 + u) & LTP_MASK] = (i16)v[u];
        }
    }
    for (int k = 0; k <= MAX_SHAPE_LPC_ORDER; k++) st.sAR_shp[k] = P.sAR_shp[k];
    st.sLTP_shp_buf_idx = P.sLTP_shp_buf_idx; st.sLF_AR_shp_Q12 = P.sLF_AR_shp_Q12; st.sLF_MA_shp_Q12 = P.sLF_MA_shp_Q12;
    st.sHarmHP_Q2 = P.sHarmHP_Q2; st.lagPrev = P.lagPrev;
    out.status = OPUSGPU_OK;
}

}  // namespace ca

using namespace ca;

extern "C" int opusgpu_silk_prefilter_batch(const opusgpu_prefilter_in *d_in, opusgpu_prefilter_state *d_state, opusgpu_prefilter_out *d_out, int n,
                                            void *stream)
{
    if (n < 0) return OPUSGPU_BAD_ARG;
    if (n == 0) return OPUSGPU_OK;
    if (!d_in || !d_state || !d_out) return OPUSGPU_BAD_ARG;
    int *bad = opusgpu_bad_record_counter();
    if (!bad) return OPUSGPU_ALLOC_FAIL;
    hipLaunchKernelGGL(silk_prefilter_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, d_in, d_state, d_out, n, bad);
    return opusgpu_check_launch();
}

extern "C" int opusgpu_silk_noise_shape_analysis_batch(const opusgpu_noise_shape_in *d_in, opusgpu_noise_shape_out *d_out, int n, void *stream)
{
    if (n < 0) return OPUSGPU_BAD_ARG;
    if (n == 0) return OPUSGPU_OK;
    if (!d_in || !d_out) return OPUSGPU_BAD_ARG;
    int *bad = opusgpu_bad_record_counter();
    if (!bad) return OPUSGPU_ALLOC_FAIL;
    hipLaunchKernelGGL(silk_noise_shape_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, d_in, d_out, n, bad);
    return opusgpu_check_launch();
}

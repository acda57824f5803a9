// silk_vad_kernels.hip -- batched silk_VAD_GetSA_Q8_c (opus-fix/silk/VAD.c:82-312), one lane per frame; the arithmetic lives in
// silk_vad_dev.h. The filter-bank buffer (5/4 frame_length samples) of each of the wavefront's frames lives in LDS [sample][lane].
#include <string.h>
#include "silk_vad_dev.h"
#include "opusgpu_internal.h"
#include "../../include/opusgpu_silk.h"
#include "silk_validate.h"

namespace ca {

struct VadCol {
    i16 *p;
    __device__ __forceinline__ i16 &operator[](int k) const { return p[k * 64]; }
    __device__ __forceinline__ VadCol operator+(int o) const { VadCol r; r.p = p + o * 64; return r; }
};

__global__ __launch_bounds__(64) void silk_vad_kernel(const opusgpu_vad_in *__restrict__ recs, opusgpu_vad_state *__restrict__ states,
                                                      opusgpu_vad_out *__restrict__ outs, int n_rec, int *__restrict__ bad_records)
{
    __shared__ i16 x_s[(OPUSGPU_SILK_MAX_FRAME * 5 / 4) * 64];
    const int r = blockIdx.x * 64 + threadIdx.x;
    if (r >= n_rec) return;
    const opusgpu_vad_in &in = recs[r];
    opusgpu_vad_state &st = states[r];
    opusgpu_vad_out o;
    memset(&o, 0, sizeof(o));
    if (!vad_record_ok(in, st)) {
        o.status = OPUSGPU_BAD_ARG;
        outs[r] = o;
        atomicAdd(bad_records, 1);
        return;
    }
    VadState V;
    for (int k = 0; k < 2; k++) { V.AnaState[k] = st.AnaState[k]; V.AnaState1[k] = st.AnaState1[k]; V.AnaState2[k] = st.AnaState2[k]; }
    for (int k = 0; k < 4; k++) {
        V.XnrgSubfr[k] = st.XnrgSubfr[k]; V.NrgRatioSmth_Q8[k] = st.NrgRatioSmth_Q8[k]; V.NL[k] = st.NL[k]; V.inv_NL[k] = st.inv_NL[k];
        V.NoiseLevelBias[k] = st.NoiseLevelBias[k];
    }
    V.HPstate = st.HPstate; V.counter = st.counter;
    VadOut vo;
    VadCol X;
    X.p = x_s + threadIdx.x;
    silk_VAD_GetSA_Q8_dev(V, vo, (const i16 *)in.pIn, X, in.frame_length, in.fs_kHz);
    for (int k = 0; k < 2; k++) { st.AnaState[k] = V.AnaState[k]; st.AnaState1[k] = V.AnaState1[k]; st.AnaState2[k] = V.AnaState2[k]; }
    for (int k = 0; k < 4; k++) { st.XnrgSubfr[k] = V.XnrgSubfr[k]; st.NrgRatioSmth_Q8[k] = V.NrgRatioSmth_Q8[k]; st.NL[k] = V.NL[k]; st.inv_NL[k] = V.inv_NL[k]; }
    st.HPstate = V.HPstate; st.counter = V.counter;
    o.speech_activity_Q8 = vo.speech_activity_Q8; o.input_tilt_Q15 = vo.input_tilt_Q15;
    for (int k = 0; k < 4; k++) o.input_quality_bands_Q15[k] = vo.input_quality_bands_Q15[k];
    o.status = OPUSGPU_OK;
    outs[r] = o;
}

}  // namespace ca

using namespace ca;

extern "C" int opusgpu_silk_vad_batch(const opusgpu_vad_in *d_in, opusgpu_vad_state *d_state, opusgpu_vad_out *d_out, int n, void *stream)
{
    if (n < 0) return OPUSGPU_BAD_ARG;
    if (n == 0) return OPUSGPU_OK;
    if (!d_in || !d_state || !d_out) return OPUSGPU_BAD_ARG;
    int *bad = opusgpu_bad_record_counter();
    if (!bad) return OPUSGPU_ALLOC_FAIL;
    hipLaunchKernelGGL(silk_vad_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, d_in, d_state, d_out, n, bad);
    return opusgpu_check_launch();
}

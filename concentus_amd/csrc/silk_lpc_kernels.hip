// silk_lpc_kernels.hip -- batched silk_find_LPC_FIX (opus-fix/silk/fixed/find_LPC_FIX.c:37-151), one lane per frame.
// The arithmetic lives in silk_lpc_dev.h / silk_burg_dev.h; this file stages the samples and checks the records.
//
// As in the Burg kernel, a record's samples are read many times (two Burg analyses, four residual filters): the
// wavefront copies its 64 records' samples into LDS once (16-byte loads), laid out [sample][lane], so that every later
// use is a conflict-free ds_read instead of a 2-byte load at an 832-byte stride between lanes.
#include <string.h>
#include "silk_lpc_dev.h"
#include "opusgpu_internal.h"
#include "../../include/opusgpu_silk.h"
#include "../../include/opusgpu_hooks.h"
#include "silk_validate.h"

namespace ca {

struct LpcX {
    const i16 *p;                                              // this lane's column of the [sample][lane] block
    __device__ __forceinline__ i32 operator[](int k) const { return p[k * 64]; }
    __device__ __forceinline__ LpcX operator+(int o) const { LpcX r; r.p = p + o * 64; return r; }
};

__global__ __launch_bounds__(64) void silk_find_lpc_kernel(const opusgpu_find_lpc_in *__restrict__ recs, opusgpu_find_lpc_out *__restrict__ outs,
                                                           int n_rec, int *__restrict__ bad_records)
{
    __shared__ i16 edge_s[BURG_EDGE_SLOTS * 64];               // the Burg recursion's subframe edges, [slot][lane]
    i16 xs[OPUSGPU_SILK_BURG_MAX_X];                           // private: streamed through in order, a handful of times
    const int r = blockIdx.x * 64 + threadIdx.x;
    if (r >= n_rec) return;
    const opusgpu_find_lpc_in &in = recs[r];
    opusgpu_find_lpc_out &o = outs[r];
    if (!find_lpc_record_ok(in)) {
        for (int k = 0; k < 16; k++) o.NLSF_Q15[k] = 0;
        o.NLSFInterpCoef_Q2 = 0;
        o.status = OPUSGPU_BAD_ARG;
        atomicAdd(bad_records, 1);
        return;
    }
    {
        static_assert(sizeof(opusgpu_find_lpc_in) % 16 == 0, "16-byte loads of x");
        const int nx = (in.subfr_length + in.predictLPCOrder) * in.nb_subfr;
        const int4 *src = reinterpret_cast<const int4 *>(in.x);
        for (int k = 0; k < nx; k += 8) {
            const int4 w = src[k >> 3];
            xs[k + 0] = (i16)w.x; xs[k + 1] = (i16)(w.x >> 16); xs[k + 2] = (i16)w.y; xs[k + 3] = (i16)(w.y >> 16);
            xs[k + 4] = (i16)w.z; xs[k + 5] = (i16)(w.z >> 16); xs[k + 6] = (i16)w.w; xs[k + 7] = (i16)(w.w >> 16);
        }
    }
    const int L = in.subfr_length + in.predictLPCOrder;
    BurgEdgesCol e;
    e.p = edge_s + threadIdx.x;
    e.stage((const i16 *)xs, L, in.nb_subfr);
    i16 prev[16], nlsf[16];
    for (int k = 0; k < 16; k++) { prev[k] = in.prev_NLSFq_Q15[k]; nlsf[k] = 0; }
    const int interp = silk_find_LPC_dev((const i16 *)xs, e, in.minInvGain_Q30, in.subfr_length, in.nb_subfr, in.predictLPCOrder, in.useInterpolatedNLSFs,
                                         in.first_frame_after_reset, prev, nlsf);
    for (int k = 0; k < 16; k++) o.NLSF_Q15[k] = nlsf[k];
    o.NLSFInterpCoef_Q2 = interp;
    o.status = OPUSGPU_OK;
}

}  // namespace ca

using namespace ca;

extern "C" int opusgpu_silk_find_lpc_batch(const opusgpu_find_lpc_in *d_in, opusgpu_find_lpc_out *d_out, int n, void *stream)
{
    if (n < 0) return OPUSGPU_BAD_ARG;
    if (n == 0) return OPUSGPU_OK;
    if (!d_in || !d_out) return OPUSGPU_BAD_ARG;
    int *bad = opusgpu_bad_record_counter();
    if (!bad) return OPUSGPU_ALLOC_FAIL;
    hipLaunchKernelGGL(silk_find_lpc_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, d_in, d_out, n, bad);
    return opusgpu_check_launch();
}

// Per-call hook with the reference's own argument list: silk_find_LPC_FIX(psEncC, NLSF_Q15, x, minInvGain_Q30)
// (silk/fixed/main_FIX.h; called at silk/fixed/find_pred_coefs_FIX.c:136). psEncC is the reference's silk_encoder_state
// (x86-64 layout): the fields read and the one written are reached at the offsets of include/opusgpu_hooks.h.
static int rd_int(const void *base, int off) { int v; memcpy(&v, (const char *)base + off, sizeof(v)); return v; }

extern "C" void opusgpu_silk_find_LPC_FIX(void *psEncC, int16_t NLSF_Q15[], const int16_t x[], const int32_t minInvGain_Q30)
{
    if (!psEncC || !NLSF_Q15 || !x) { opusgpu_set_last_error(OPUSGPU_BAD_ARG); return; }
    opusgpu_find_lpc_in h_in;
    memset(&h_in, 0, sizeof(h_in));
    h_in.minInvGain_Q30 = minInvGain_Q30;
    h_in.subfr_length = rd_int(psEncC, OPUSGPU_REF_OFF_SUBFR_LENGTH);
    h_in.nb_subfr = rd_int(psEncC, OPUSGPU_REF_OFF_NB_SUBFR);
    h_in.predictLPCOrder = rd_int(psEncC, OPUSGPU_REF_OFF_PREDICT_LPC_ORDER);
    h_in.useInterpolatedNLSFs = rd_int(psEncC, OPUSGPU_REF_OFF_USE_INTERPOLATED_NLSFS);
    h_in.first_frame_after_reset = rd_int(psEncC, OPUSGPU_REF_OFF_FIRST_FRAME_AFTER_RESET);
    memcpy(h_in.prev_NLSFq_Q15, (const char *)psEncC + OPUSGPU_REF_OFF_PREV_NLSFQ_Q15, sizeof(h_in.prev_NLSFq_Q15));
    const long nx = (long)(h_in.subfr_length + h_in.predictLPCOrder) * h_in.nb_subfr;
    if (h_in.nb_subfr < 1 || h_in.subfr_length < 1 || h_in.predictLPCOrder < 1 || nx > OPUSGPU_SILK_BURG_MAX_X) {
        opusgpu_set_last_error(OPUSGPU_BAD_ARG);
        return;
    }
    memcpy(h_in.x, x, sizeof(int16_t) * (size_t)nx);
    opusgpu_find_lpc_in *d_in = nullptr;
    opusgpu_find_lpc_out *d_out = nullptr, h_out;
    if (hipMalloc(&d_in, sizeof(h_in)) != hipSuccess || hipMalloc(&d_out, sizeof(h_out)) != hipSuccess) {
        opusgpu_set_last_error(OPUSGPU_ALLOC_FAIL);
        if (d_in) (void)hipFree(d_in);
        return;
    }
    int rc = hipMemcpy(d_in, &h_in, sizeof(h_in), hipMemcpyHostToDevice) == hipSuccess ? OPUSGPU_OK : OPUSGPU_INTERNAL_ERROR;
    OpusgpuHookBadScope bad;                 // rejected records count into this thread's counter, not the device's shared one
    if (rc == OPUSGPU_OK) rc = bad.rc;
    if (rc == OPUSGPU_OK) rc = opusgpu_silk_find_lpc_batch(d_in, d_out, 1, nullptr);
    if (rc == OPUSGPU_OK && hipMemcpy(&h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost) != hipSuccess) rc = OPUSGPU_INTERNAL_ERROR;
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    if (rc == OPUSGPU_OK && h_out.status != OPUSGPU_OK) rc = h_out.status;
    opusgpu_set_last_error(rc);
    if (rc != OPUSGPU_OK) return;
    memcpy(NLSF_Q15, h_out.NLSF_Q15, sizeof(int16_t) * (size_t)h_in.predictLPCOrder);
    *((int8_t *)psEncC + OPUSGPU_REF_OFF_INDICES + OPUSGPU_REF_OFF_NLSF_INTERP_COEF_Q2) = (int8_t)h_out.NLSFInterpCoef_Q2;
}

// silk_pred_kernels.hip -- batched silk_find_pred_coefs_FIX (opus-fix/silk/fixed/find_pred_coefs_FIX.c:35-148), one lane per frame.
// The arithmetic lives in silk_pred_dev.h; this file checks the records and provides LPC_in_pre's storage: the wavefront's 64
// LPC_in_pre signals live in LDS laid out [sample][lane] (the Burg analyses and the residual filters read every sample many
// times); res_pitch and x are read from the records where they lie.
#include <string.h>
#include "silk_pred_dev.h"
#include "opusgpu_internal.h"
#include "../../include/opusgpu_silk.h"
#include "silk_validate.h"

namespace ca {

struct PreCol {                                                // this lane's column of the [sample][lane] block
    i16 *p;
    __device__ __forceinline__ i16 &operator[](int k) const { return p[k * 64]; }
    __device__ __forceinline__ PreCol operator+(int o) const { PreCol r; r.p = p + o * 64; return r; }
};

__global__ __launch_bounds__(64) void silk_find_pred_coefs_kernel(const opusgpu_find_pred_coefs_in *__restrict__ recs,
                                                                  opusgpu_find_pred_coefs_out *__restrict__ outs, int n_rec,
                                                                  int *__restrict__ bad_records)
{
    __shared__ i16 pre_s[OPUSGPU_SILK_BURG_MAX_X * 64];
    const int r = blockIdx.x * 64 + threadIdx.x;
    if (r >= n_rec) return;
    const opusgpu_find_pred_coefs_in &in = recs[r];
    opusgpu_find_pred_coefs_out &out = outs[r];
    if (!find_pred_coefs_record_ok(in)) {
        memset(&out, 0, sizeof(out));
        out.status = OPUSGPU_BAD_ARG;
        atomicAdd(bad_records, 1);
        return;
    }
    PredCoefsCfg c;
    for (int k = 0; k < 4; k++) { c.Gains_Q16[k] = in.Gains_Q16[k]; c.pitchL[k] = in.pitchL[k]; }
    for (int k = 0; k < SILK_MAX_LPC; k++) c.prev_NLSFq_Q15[k] = in.prev_NLSFq_Q15[k];
    c.nb_subfr = in.nb_subfr; c.subfr_length = in.subfr_length; c.predictLPCOrder = in.predictLPCOrder; c.ltp_mem_length = in.ltp_mem_length;
    c.signalType = in.signalType; c.condCoding = in.condCoding; c.first_frame_after_reset = in.first_frame_after_reset;
    c.useInterpolatedNLSFs = in.useInterpolatedNLSFs; c.speech_activity_Q8 = in.speech_activity_Q8;
    c.NLSF_MSVQ_Survivors = in.NLSF_MSVQ_Survivors; c.mu_LTP_Q9 = in.mu_LTP_Q9; c.LTPQuantLowComplexity = in.LTPQuantLowComplexity;
    c.sum_log_gain_Q7 = in.sum_log_gain_Q7; c.coding_quality_Q14 = in.coding_quality_Q14; c.PacketLoss_perc = in.PacketLoss_perc;
    c.nFramesPerPacket = in.nFramesPerPacket;
    PredCoefsOut o;
    memset(&o, 0, sizeof(o));
    PreCol pre;
    pre.p = pre_s + threadIdx.x;
    silk_find_pred_coefs_dev(c, (const i16 *)in.res_pitch, (const i16 *)in.x + in.ltp_mem_length, pre, o);
    const int order = in.predictLPCOrder, nb = in.nb_subfr;
    memset(&out, 0, sizeof(out));
    for (int k = 0; k < order; k++) { out.PredCoef_Q12[0][k] = o.PredCoef_Q12[0][k]; out.PredCoef_Q12[1][k] = o.PredCoef_Q12[1][k]; out.NLSF_Q15[k] = o.NLSF_Q15[k]; }
    for (int k = 0; k < nb * LTP_ORDER; k++) out.LTPCoef_Q14[k] = o.LTPCoef_Q14[k];
    for (int k = 0; k < nb; k++) { out.ResNrg[k] = o.ResNrg[k]; out.ResNrgQ[k] = o.ResNrgQ[k]; out.LTPIndex[k] = o.LTPIndex[k]; }
    out.LTPredCodGain_Q7 = o.LTPredCodGain_Q7; out.LTP_scale_Q14 = o.LTP_scale_Q14; out.sum_log_gain_Q7 = o.sum_log_gain_Q7;
    for (int k = 0; k <= order; k++) out.NLSFIndices[k] = o.NLSFIndices[k];
    out.NLSFInterpCoef_Q2 = (i8)o.NLSFInterpCoef_Q2; out.PERIndex = (i8)o.PERIndex; out.LTP_scaleIndex = (i8)o.LTP_scaleIndex;
    out.status = OPUSGPU_OK;
}

}  // namespace ca

using namespace ca;

extern "C" int opusgpu_silk_find_pred_coefs_batch(const opusgpu_find_pred_coefs_in *d_in, opusgpu_find_pred_coefs_out *d_out, int n, void *stream)
{
    if (n < 0) return OPUSGPU_BAD_ARG;
    if (n == 0) return OPUSGPU_OK;
    if (!d_in || !d_out) return OPUSGPU_BAD_ARG;
    int *bad = opusgpu_bad_record_counter();
    if (!bad) return OPUSGPU_ALLOC_FAIL;
    hipLaunchKernelGGL(silk_find_pred_coefs_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, d_in, d_out, n, bad);
    return opusgpu_check_launch();
}

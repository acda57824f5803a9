// silk_pred_kernels.hip -- batched silk_find_pred_coefs_FIX (opus-fix/silk/fixed/find_pred_coefs_FIX.c:35-148), one lane per frame.
// The arithmetic lives in silk_pred_dev.h; this file checks the records and provides LPC_in_pre's storage: private memory (384
// samples, written once and streamed through a few times in order) plus the 2 x 16 edge samples of every subframe, which the Burg
// recursion revisits in dependent chains, in LDS laid out [slot][lane] -- 18 KB per workgroup, so that four workgroups (one
// wavefront per SIMD, all 65 536 frames of a full batch) are resident per CU. res_pitch and x are read from the records where they lie.
#include <string.h>
#include "silk_pred_dev.h"
#include "silk_gains_dev.h"
#include "opusgpu_internal.h"
#include "../../include/opusgpu_silk.h"
#include "../../include/opusgpu_hooks.h"
#include "silk_validate.h"

namespace ca {

__global__ __launch_bounds__(64) void silk_find_pred_coefs_kernel(const opusgpu_find_pred_coefs_in *__restrict__ recs,
                                                                  opusgpu_find_pred_coefs_out *__restrict__ outs, int n_rec,
                                                                  int *__restrict__ bad_records)
{
    __shared__ __attribute__((aligned(16))) i16 edge_s[BURG_EDGE_SLOTS * 64];
    __shared__ NlsfTablesLds tables;
    __shared__ NlsfEncTables enc;
    nlsf_stage_tables(tables, threadIdx.x, 64);
    nlsf_stage_enc_tables(enc, threadIdx.x, 64);
    __syncthreads();
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
    i16 pre[OPUSGPU_SILK_BURG_MAX_X];
    BurgEdgesCol e;
    e.p = edge_s + threadIdx.x;
    // (private copies of res_pitch / x, fetched eight samples per access, were measured: 2.20 -> 2.48 ms -- the LTP analysis' windows
    // are served by the L1 as they lie in the record)
    silk_find_pred_coefs_dev(c, (const i16 *)in.res_pitch, (const i16 *)in.x + in.ltp_mem_length, (i16 *)pre, e, o, &tables, &enc);
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

__global__ __launch_bounds__(64) void silk_process_gains_kernel(const opusgpu_process_gains_in *__restrict__ recs,
                                                                opusgpu_process_gains_out *__restrict__ outs, int n_rec,
                                                                int *__restrict__ bad_records)
{
    const int r = blockIdx.x * 64 + threadIdx.x;
    if (r >= n_rec) return;
    const opusgpu_process_gains_in in = recs[r];
    opusgpu_process_gains_out o;
    memset(&o, 0, sizeof(o));
    if (!process_gains_record_ok(in)) {
        o.status = OPUSGPU_BAD_ARG;
        outs[r] = o;
        atomicAdd(bad_records, 1);
        return;
    }
    ProcessGainsIO g;
    for (int k = 0; k < 4; k++) { g.Gains_Q16[k] = in.Gains_Q16[k]; g.GainsUnq_Q16[k] = 0; g.ResNrg[k] = in.ResNrg[k]; g.ResNrgQ[k] = in.ResNrgQ[k]; g.GainsIndices[k] = 0; }
    g.LastGainIndex = in.LastGainIndex; g.lastGainIndexPrev = 0; g.quantOffsetType = in.quantOffsetType; g.Lambda_Q10 = 0;
    silk_process_gains_dev(g, in.signalType, in.nb_subfr, in.subfr_length, in.LTPredCodGain_Q7, in.SNR_dB_Q7, in.condCoding,
                           in.input_tilt_Q15, in.nStatesDelayedDecision, in.speech_activity_Q8, in.input_quality_Q14, in.coding_quality_Q14);
    for (int k = 0; k < in.nb_subfr; k++) { o.Gains_Q16[k] = g.Gains_Q16[k]; o.GainsUnq_Q16[k] = g.GainsUnq_Q16[k]; o.GainsIndices[k] = g.GainsIndices[k]; }
    o.Lambda_Q10 = g.Lambda_Q10; o.LastGainIndex = g.LastGainIndex; o.lastGainIndexPrev = g.lastGainIndexPrev; o.quantOffsetType = g.quantOffsetType;
    o.status = OPUSGPU_OK;
    outs[r] = o;
}

}  // namespace ca

using namespace ca;

extern "C" int opusgpu_silk_process_gains_batch(const opusgpu_process_gains_in *d_in, opusgpu_process_gains_out *d_out, int n, void *stream)
{
    if (n < 0) return OPUSGPU_BAD_ARG;
    if (n == 0) return OPUSGPU_OK;
    if (!d_in || !d_out) return OPUSGPU_BAD_ARG;
    int *bad = opusgpu_bad_record_counter();
    if (!bad) return OPUSGPU_ALLOC_FAIL;
    hipLaunchKernelGGL(silk_process_gains_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, d_in, d_out, n, bad);
    return opusgpu_check_launch();
}

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

// Per-call hook with the reference's own argument list: silk_find_pred_coefs_FIX(psEnc, psEncCtrl, res_pitch, x, condCoding)
// (silk/fixed/main_FIX.h; called from silk_encode_frame_FIX). psEnc / psEncCtrl are the reference's silk_encoder_state_FIX /
// silk_encoder_control_FIX (x86-64 layout): the fields read and written are reached at the offsets of include/opusgpu_hooks.h.
// x must be preceded by ltp_mem_length samples (it points into psEnc->x_buf, encode_frame_FIX.c), res_pitch holds
// ltp_mem_length + frame_length samples.
static int rd_int(const void *base, int off) { int v; memcpy(&v, (const char *)base + off, sizeof(v)); return v; }
static void wr_int(void *base, int off, int v) { memcpy((char *)base + off, &v, sizeof(v)); }

extern "C" void opusgpu_silk_find_pred_coefs_FIX(void *psEnc, void *psEncCtrl, const int16_t res_pitch[], const int16_t x[], int condCoding)
{
    if (!psEnc || !psEncCtrl || !res_pitch || !x) { opusgpu_set_last_error(OPUSGPU_BAD_ARG); return; }
    static opusgpu_find_pred_coefs_in h_in_zero;
    opusgpu_find_pred_coefs_in *h_in = (opusgpu_find_pred_coefs_in *)malloc(sizeof(*h_in));
    opusgpu_find_pred_coefs_out h_out;
    if (!h_in) { opusgpu_set_last_error(OPUSGPU_ALLOC_FAIL); return; }
    *h_in = h_in_zero;
    char *sCmn = (char *)psEnc + OPUSGPU_REF_OFF_FIX_SCMN;
    char *ind = sCmn + OPUSGPU_REF_OFF_INDICES;
    h_in->nb_subfr = rd_int(sCmn, OPUSGPU_REF_OFF_NB_SUBFR);
    h_in->subfr_length = rd_int(sCmn, OPUSGPU_REF_OFF_SUBFR_LENGTH);
    h_in->predictLPCOrder = rd_int(sCmn, OPUSGPU_REF_OFF_PREDICT_LPC_ORDER);
    h_in->ltp_mem_length = rd_int(sCmn, OPUSGPU_REF_OFF_LTP_MEM_LENGTH);
    h_in->signalType = (int8_t)ind[OPUSGPU_REF_OFF_SIGNAL_TYPE];
    h_in->condCoding = condCoding;
    h_in->first_frame_after_reset = rd_int(sCmn, OPUSGPU_REF_OFF_FIRST_FRAME_AFTER_RESET);
    h_in->useInterpolatedNLSFs = rd_int(sCmn, OPUSGPU_REF_OFF_USE_INTERPOLATED_NLSFS);
    h_in->speech_activity_Q8 = rd_int(sCmn, OPUSGPU_REF_OFF_SPEECH_ACTIVITY_Q8);
    h_in->NLSF_MSVQ_Survivors = rd_int(sCmn, OPUSGPU_REF_OFF_NLSF_MSVQ_SURVIVORS);
    h_in->mu_LTP_Q9 = rd_int(sCmn, OPUSGPU_REF_OFF_MU_LTP_Q9);
    h_in->LTPQuantLowComplexity = rd_int(sCmn, OPUSGPU_REF_OFF_LTP_QUANT_LOW_COMPLEXITY);
    h_in->sum_log_gain_Q7 = rd_int(sCmn, OPUSGPU_REF_OFF_SUM_LOG_GAIN_Q7);
    h_in->PacketLoss_perc = rd_int(sCmn, OPUSGPU_REF_OFF_PACKET_LOSS_PERC);
    h_in->nFramesPerPacket = rd_int(sCmn, OPUSGPU_REF_OFF_N_FRAMES_PER_PACKET);
    h_in->coding_quality_Q14 = rd_int(psEncCtrl, OPUSGPU_REF_OFF_CTRL_CODING_QUALITY_Q14);
    memcpy(h_in->Gains_Q16, (char *)psEncCtrl + OPUSGPU_REF_OFF_CTRL_GAINS_Q16, sizeof(h_in->Gains_Q16));
    memcpy(h_in->pitchL, (char *)psEncCtrl + OPUSGPU_REF_OFF_CTRL_PITCHL, sizeof(h_in->pitchL));
    memcpy(h_in->prev_NLSFq_Q15, sCmn + OPUSGPU_REF_OFF_PREV_NLSFQ_Q15, sizeof(h_in->prev_NLSFq_Q15));
    const int nb = h_in->nb_subfr, L = h_in->subfr_length, D = h_in->predictLPCOrder, ltp = h_in->ltp_mem_length;
    if ((nb != 2 && nb != 4) || L < 1 || L > 80 || (D != 10 && D != 16) || ltp < D || ltp > OPUSGPU_SILK_MAX_LTP_MEM) {
        free(h_in);
        opusgpu_set_last_error(OPUSGPU_BAD_ARG);
        return;
    }
    memcpy(h_in->res_pitch, res_pitch, sizeof(int16_t) * (size_t)(ltp + nb * L));
    memcpy(h_in->x, x - ltp, sizeof(int16_t) * (size_t)(ltp + nb * L));
    opusgpu_find_pred_coefs_in *d_in = nullptr;
    opusgpu_find_pred_coefs_out *d_out = nullptr;
    int rc = OPUSGPU_OK;
    if (hipMalloc(&d_in, sizeof(*h_in)) != hipSuccess || hipMalloc(&d_out, sizeof(h_out)) != hipSuccess) rc = OPUSGPU_ALLOC_FAIL;
    if (rc == OPUSGPU_OK && hipMemcpy(d_in, h_in, sizeof(*h_in), hipMemcpyHostToDevice) != hipSuccess) rc = OPUSGPU_INTERNAL_ERROR;
    OpusgpuHookBadScope bad;                 // rejected records count into this thread's counter, not the device's shared one
    if (rc == OPUSGPU_OK) rc = bad.rc;
    if (rc == OPUSGPU_OK) rc = opusgpu_silk_find_pred_coefs_batch(d_in, d_out, 1, nullptr);
    if (rc == OPUSGPU_OK && hipMemcpy(&h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost) != hipSuccess) rc = OPUSGPU_INTERNAL_ERROR;
    if (d_in) (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    const int voiced = h_in->signalType == 2;
    free(h_in);
    if (rc == OPUSGPU_OK && h_out.status != OPUSGPU_OK) rc = h_out.status;
    opusgpu_set_last_error(rc);
    if (rc != OPUSGPU_OK) return;
    char *ctl = (char *)psEncCtrl;
    memcpy(ctl + OPUSGPU_REF_OFF_CTRL_PRED_COEF_Q12, h_out.PredCoef_Q12[0], sizeof(int16_t) * (size_t)D);
    memcpy(ctl + OPUSGPU_REF_OFF_CTRL_PRED_COEF_Q12 + 2 * OPUSGPU_SILK_MAX_ORDER, h_out.PredCoef_Q12[1], sizeof(int16_t) * (size_t)D);
    memcpy(ctl + OPUSGPU_REF_OFF_CTRL_LTP_COEF_Q14, h_out.LTPCoef_Q14, sizeof(int16_t) * (size_t)(nb * 5));
    wr_int(ctl, OPUSGPU_REF_OFF_CTRL_LTP_RED_COD_GAIN_Q7, h_out.LTPredCodGain_Q7);
    for (int k = 0; k < nb; k++) {
        wr_int(ctl, OPUSGPU_REF_OFF_CTRL_RES_NRG + 4 * k, h_out.ResNrg[k]);
        wr_int(ctl, OPUSGPU_REF_OFF_CTRL_RES_NRG_Q + 4 * k, h_out.ResNrgQ[k]);
    }
    wr_int(sCmn, OPUSGPU_REF_OFF_SUM_LOG_GAIN_Q7, h_out.sum_log_gain_Q7);
    memcpy(sCmn + OPUSGPU_REF_OFF_PREV_NLSFQ_Q15, h_out.NLSF_Q15, sizeof(int16_t) * OPUSGPU_SILK_MAX_ORDER);
    memcpy(ind + OPUSGPU_REF_OFF_NLSF_INDICES, h_out.NLSFIndices, (size_t)D + 1);
    ind[OPUSGPU_REF_OFF_NLSF_INTERP_COEF_Q2] = h_out.NLSFInterpCoef_Q2;
    if (voiced) {
        wr_int(ctl, OPUSGPU_REF_OFF_CTRL_LTP_SCALE_Q14, h_out.LTP_scale_Q14);
        memcpy(ind + OPUSGPU_REF_OFF_LTP_INDEX, h_out.LTPIndex, (size_t)nb);
        ind[OPUSGPU_REF_OFF_PER_INDEX] = h_out.PERIndex;
        ind[OPUSGPU_REF_OFF_LTP_SCALE_INDEX] = h_out.LTP_scaleIndex;
    }
}

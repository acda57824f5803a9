// silk_frame_hooks.hip -- per-call hooks, with the reference's own argument lists, for the four analysis calls of
// silk_encode_frame_FIX that have no record-level hook elsewhere: silk_find_pitch_lags_FIX, silk_noise_shape_analysis_FIX,
// silk_process_gains_FIX, silk_prefilter_FIX (include/opusgpu_hooks.h). Each builds the function's record from the reference's
// silk_encoder_state_FIX / silk_encoder_control_FIX (x86-64 layout, offsets pinned by tests/test_hooks_layout.py), runs the
// batched kernel on one record and writes back exactly the fields the reference function writes. Plumbing / parity only.
#include <stdlib.h>
#include <string.h>
#include <hip/hip_runtime.h>
#include "opusgpu_internal.h"
#include "../../include/opusgpu_silk.h"
#include "../../include/opusgpu_hooks.h"

static int rd_int(const void *base, int off) { int v; memcpy(&v, (const char *)base + off, sizeof(v)); return v; }
static void wr_int(void *base, int off, int v) { memcpy((char *)base + off, &v, sizeof(v)); }

// one record through a batched entry point: host in -> device -> host out (and an optional in/out state record)
template <class In, class Out, class State, class Launch>
static int run_record(const In *h_in, Out *h_out, State *h_state, Launch launch)
{
    In *d_in = nullptr;
    Out *d_out = nullptr;
    State *d_state = nullptr;
    OpusgpuHookBadScope bad;                 // rejected records count into this thread's counter, not the device's shared one
    int rc = bad.rc;
    if (rc != OPUSGPU_OK) return rc;
    if (hipMalloc(&d_in, sizeof(In)) != hipSuccess || hipMalloc(&d_out, sizeof(Out)) != hipSuccess ||
        (h_state && hipMalloc(&d_state, sizeof(State)) != hipSuccess))
        rc = OPUSGPU_ALLOC_FAIL;
    if (rc == OPUSGPU_OK && hipMemcpy(d_in, h_in, sizeof(In), hipMemcpyHostToDevice) != hipSuccess) rc = OPUSGPU_INTERNAL_ERROR;
    if (rc == OPUSGPU_OK && h_state && hipMemcpy(d_state, h_state, sizeof(State), hipMemcpyHostToDevice) != hipSuccess) rc = OPUSGPU_INTERNAL_ERROR;
    if (rc == OPUSGPU_OK) rc = launch(d_in, d_state, d_out);
    if (rc == OPUSGPU_OK && hipMemcpy(h_out, d_out, sizeof(Out), hipMemcpyDeviceToHost) != hipSuccess) rc = OPUSGPU_INTERNAL_ERROR;
    if (rc == OPUSGPU_OK && h_out->status != OPUSGPU_OK) rc = h_out->status;
    if (rc == OPUSGPU_OK && h_state && hipMemcpy(h_state, d_state, sizeof(State), hipMemcpyDeviceToHost) != hipSuccess) rc = OPUSGPU_INTERNAL_ERROR;
    if (d_in) (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    if (d_state) (void)hipFree(d_state);
    return rc;
}

struct NoState { int unused; };

extern "C" void opusgpu_silk_find_pitch_lags_FIX(void *psEnc, void *psEncCtrl, int16_t res[], const int16_t x[], int arch)
{
    (void)arch;
    if (!psEnc || !psEncCtrl || !res || !x) { opusgpu_set_last_error(OPUSGPU_BAD_ARG); return; }
    char *sCmn = (char *)psEnc + OPUSGPU_REF_OFF_FIX_SCMN, *ind = sCmn + OPUSGPU_REF_OFF_INDICES;
    opusgpu_find_pitch_lags_in *in = (opusgpu_find_pitch_lags_in *)calloc(1, sizeof(*in));
    opusgpu_find_pitch_lags_out *out = (opusgpu_find_pitch_lags_out *)calloc(1, sizeof(*out));
    if (!in || !out) { free(in); free(out); opusgpu_set_last_error(OPUSGPU_ALLOC_FAIL); return; }
    in->fs_kHz = rd_int(sCmn, OPUSGPU_REF_OFF_FS_KHZ); in->nb_subfr = rd_int(sCmn, OPUSGPU_REF_OFF_NB_SUBFR);
    in->frame_length = rd_int(sCmn, OPUSGPU_REF_OFF_FRAME_LENGTH); in->ltp_mem_length = rd_int(sCmn, OPUSGPU_REF_OFF_LTP_MEM_LENGTH);
    in->la_pitch = rd_int(sCmn, OPUSGPU_REF_OFF_LA_PITCH); in->pitch_LPC_win_length = rd_int(sCmn, OPUSGPU_REF_OFF_PITCH_LPC_WIN_LENGTH);
    in->pitchEstimationLPCOrder = rd_int(sCmn, OPUSGPU_REF_OFF_PITCH_EST_LPC_ORDER);
    in->pitchEstimationComplexity = rd_int(sCmn, OPUSGPU_REF_OFF_PITCH_EST_COMPLEXITY);
    in->pitchEstimationThreshold_Q16 = rd_int(sCmn, OPUSGPU_REF_OFF_PITCH_EST_THRESHOLD_Q16);
    in->signalType = (int8_t)ind[OPUSGPU_REF_OFF_SIGNAL_TYPE];
    in->first_frame_after_reset = rd_int(sCmn, OPUSGPU_REF_OFF_FIRST_FRAME_AFTER_RESET);
    in->speech_activity_Q8 = rd_int(sCmn, OPUSGPU_REF_OFF_SPEECH_ACTIVITY_Q8);
    in->prevSignalType = (int8_t)sCmn[OPUSGPU_REF_OFF_PREV_SIGNAL_TYPE];
    in->input_tilt_Q15 = rd_int(sCmn, OPUSGPU_REF_OFF_INPUT_TILT_Q15); in->prevLag = rd_int(sCmn, OPUSGPU_REF_OFF_PREV_LAG);
    in->LTPCorr_Q15 = rd_int(psEnc, OPUSGPU_REF_OFF_FIX_LTPCORR_Q15);
    const long buf_len = (long)in->la_pitch + in->frame_length + in->ltp_mem_length;
    if (in->la_pitch < 0 || in->frame_length < 0 || in->ltp_mem_length < 0 || buf_len > OPUSGPU_SILK_PITCH_BUF) {
        free(in); free(out);
        opusgpu_set_last_error(OPUSGPU_BAD_ARG);
        return;
    }
    memcpy(in->x_buf, x - in->ltp_mem_length, sizeof(int16_t) * (size_t)buf_len);
    const int rc = run_record(in, out, (NoState *)nullptr, [](const opusgpu_find_pitch_lags_in *i, NoState *, opusgpu_find_pitch_lags_out *o) {
        return opusgpu_silk_find_pitch_lags_batch(i, o, 1, nullptr);
    });
    opusgpu_set_last_error(rc);
    if (rc == OPUSGPU_OK) {
        memcpy(res, out->res, sizeof(int16_t) * (size_t)buf_len);
        for (int k = 0; k < 4; k++) wr_int(psEncCtrl, OPUSGPU_REF_OFF_CTRL_PITCHL + 4 * k, k < in->nb_subfr || in->signalType == 0 || in->first_frame_after_reset ? out->pitchL[k] : rd_int(psEncCtrl, OPUSGPU_REF_OFF_CTRL_PITCHL + 4 * k));
        wr_int(psEncCtrl, OPUSGPU_REF_OFF_CTRL_PRED_GAIN_Q16, out->predGain_Q16);
        const int16_t lag = (int16_t)out->lagIndex;
        memcpy(ind + OPUSGPU_REF_OFF_LAG_INDEX, &lag, sizeof(lag));
        ind[OPUSGPU_REF_OFF_CONTOUR_INDEX] = (char)out->contourIndex;
        ind[OPUSGPU_REF_OFF_SIGNAL_TYPE] = (char)out->signalType;
        wr_int(psEnc, OPUSGPU_REF_OFF_FIX_LTPCORR_Q15, out->LTPCorr_Q15);
    }
    free(in); free(out);
}

extern "C" void opusgpu_silk_noise_shape_analysis_FIX(void *psEnc, void *psEncCtrl, const int16_t *pitch_res, const int16_t *x, int arch)
{
    (void)arch;
    if (!psEnc || !psEncCtrl || !pitch_res || !x) { opusgpu_set_last_error(OPUSGPU_BAD_ARG); return; }
    char *sCmn = (char *)psEnc + OPUSGPU_REF_OFF_FIX_SCMN, *ind = sCmn + OPUSGPU_REF_OFF_INDICES, *shp = (char *)psEnc + OPUSGPU_REF_OFF_FIX_SSHAPE;
    opusgpu_noise_shape_in *in = (opusgpu_noise_shape_in *)calloc(1, sizeof(*in));
    opusgpu_noise_shape_out *out = (opusgpu_noise_shape_out *)calloc(1, sizeof(*out));
    if (!in || !out) { free(in); free(out); opusgpu_set_last_error(OPUSGPU_ALLOC_FAIL); return; }
    in->fs_kHz = rd_int(sCmn, OPUSGPU_REF_OFF_FS_KHZ); in->nb_subfr = rd_int(sCmn, OPUSGPU_REF_OFF_NB_SUBFR);
    in->subfr_length = rd_int(sCmn, OPUSGPU_REF_OFF_SUBFR_LENGTH); in->la_shape = rd_int(sCmn, OPUSGPU_REF_OFF_LA_SHAPE);
    in->shapeWinLength = rd_int(sCmn, OPUSGPU_REF_OFF_SHAPE_WIN_LENGTH); in->shapingLPCOrder = rd_int(sCmn, OPUSGPU_REF_OFF_SHAPING_LPC_ORDER);
    in->warping_Q16 = rd_int(sCmn, OPUSGPU_REF_OFF_WARPING_Q16); in->SNR_dB_Q7 = rd_int(sCmn, OPUSGPU_REF_OFF_SNR_DB_Q7);
    in->useCBR = rd_int(sCmn, OPUSGPU_REF_OFF_USE_CBR); in->speech_activity_Q8 = rd_int(sCmn, OPUSGPU_REF_OFF_SPEECH_ACTIVITY_Q8);
    in->signalType = (int8_t)ind[OPUSGPU_REF_OFF_SIGNAL_TYPE]; in->LTPCorr_Q15 = rd_int(psEnc, OPUSGPU_REF_OFF_FIX_LTPCORR_Q15);
    in->input_quality_bands_Q15[0] = rd_int(sCmn, OPUSGPU_REF_OFF_INPUT_QUALITY_BANDS_Q15);
    in->input_quality_bands_Q15[1] = rd_int(sCmn, OPUSGPU_REF_OFF_INPUT_QUALITY_BANDS_Q15 + 4);
    in->predGain_Q16 = rd_int(psEncCtrl, OPUSGPU_REF_OFF_CTRL_PRED_GAIN_Q16);
    for (int k = 0; k < 4; k++) in->pitchL[k] = rd_int(psEncCtrl, OPUSGPU_REF_OFF_CTRL_PITCHL + 4 * k);
    in->HarmBoost_smth_Q16 = rd_int(shp, OPUSGPU_REF_OFF_SHAPE_HARM_BOOST_SMTH_Q16);
    in->HarmShapeGain_smth_Q16 = rd_int(shp, OPUSGPU_REF_OFF_SHAPE_HARM_SHAPE_GAIN_SMTH_Q16);
    in->Tilt_smth_Q16 = rd_int(shp, OPUSGPU_REF_OFF_SHAPE_TILT_SMTH_Q16);
    const long fl = (long)in->nb_subfr * in->subfr_length;
    if (in->nb_subfr < 1 || in->subfr_length < 1 || in->la_shape < 0 || in->la_shape > OPUSGPU_SILK_MAX_LA_SHAPE || fl > OPUSGPU_SILK_MAX_FRAME) {
        free(in); free(out);
        opusgpu_set_last_error(OPUSGPU_BAD_ARG);
        return;
    }
    memcpy(in->x, x - in->la_shape, sizeof(int16_t) * (size_t)(fl + 2 * in->la_shape));
    memcpy(in->pitch_res, pitch_res, sizeof(int16_t) * (size_t)fl);
    const int rc = run_record(in, out, (NoState *)nullptr, [](const opusgpu_noise_shape_in *i, NoState *, opusgpu_noise_shape_out *o) {
        return opusgpu_silk_noise_shape_analysis_batch(i, o, 1, nullptr);
    });
    opusgpu_set_last_error(rc);
    if (rc == OPUSGPU_OK) {
        char *ctl = (char *)psEncCtrl;
        const int nb = in->nb_subfr, D = in->shapingLPCOrder;
        for (int k = 0; k < nb; k++) {
            wr_int(ctl, OPUSGPU_REF_OFF_CTRL_GAINS_Q16 + 4 * k, out->Gains_Q16[k]);
            wr_int(ctl, OPUSGPU_REF_OFF_CTRL_GAINS_PRE_Q14 + 4 * k, out->GainsPre_Q14[k]);
            wr_int(ctl, OPUSGPU_REF_OFF_CTRL_LF_SHP_Q14 + 4 * k, out->LF_shp_Q14[k]);
            memcpy(ctl + OPUSGPU_REF_OFF_CTRL_AR1_Q13 + 32 * k, &out->AR1_Q13[16 * k], sizeof(int16_t) * (size_t)D);
            memcpy(ctl + OPUSGPU_REF_OFF_CTRL_AR2_Q13 + 32 * k, &out->AR2_Q13[16 * k], sizeof(int16_t) * (size_t)D);
        }
        for (int k = 0; k < 4; k++) {
            wr_int(ctl, OPUSGPU_REF_OFF_CTRL_HARM_BOOST_Q14 + 4 * k, out->HarmBoost_Q14[k]);
            wr_int(ctl, OPUSGPU_REF_OFF_CTRL_HARM_SHAPE_GAIN_Q14 + 4 * k, out->HarmShapeGain_Q14[k]);
            wr_int(ctl, OPUSGPU_REF_OFF_CTRL_TILT_Q14 + 4 * k, out->Tilt_Q14[k]);
        }
        wr_int(ctl, OPUSGPU_REF_OFF_CTRL_INPUT_QUALITY_Q14, out->input_quality_Q14);
        wr_int(ctl, OPUSGPU_REF_OFF_CTRL_CODING_QUALITY_Q14, out->coding_quality_Q14);
        wr_int(ctl, OPUSGPU_REF_OFF_CTRL_SPARSENESS_Q8, out->sparseness_Q8);
        ind[OPUSGPU_REF_OFF_QUANT_OFFSET_TYPE] = (char)out->quantOffsetType;
        wr_int(shp, OPUSGPU_REF_OFF_SHAPE_HARM_BOOST_SMTH_Q16, out->HarmBoost_smth_Q16);
        wr_int(shp, OPUSGPU_REF_OFF_SHAPE_HARM_SHAPE_GAIN_SMTH_Q16, out->HarmShapeGain_smth_Q16);
        wr_int(shp, OPUSGPU_REF_OFF_SHAPE_TILT_SMTH_Q16, out->Tilt_smth_Q16);
    }
    free(in); free(out);
}

extern "C" void opusgpu_silk_process_gains_FIX(void *psEnc, void *psEncCtrl, int condCoding)
{
    if (!psEnc || !psEncCtrl) { opusgpu_set_last_error(OPUSGPU_BAD_ARG); return; }
    char *sCmn = (char *)psEnc + OPUSGPU_REF_OFF_FIX_SCMN, *ind = sCmn + OPUSGPU_REF_OFF_INDICES, *shp = (char *)psEnc + OPUSGPU_REF_OFF_FIX_SSHAPE;
    char *ctl = (char *)psEncCtrl;
    opusgpu_process_gains_in in;
    opusgpu_process_gains_out out;
    memset(&in, 0, sizeof(in));
    memset(&out, 0, sizeof(out));
    for (int k = 0; k < 4; k++) {
        in.Gains_Q16[k] = rd_int(ctl, OPUSGPU_REF_OFF_CTRL_GAINS_Q16 + 4 * k);
        in.ResNrg[k] = rd_int(ctl, OPUSGPU_REF_OFF_CTRL_RES_NRG + 4 * k);
        in.ResNrgQ[k] = rd_int(ctl, OPUSGPU_REF_OFF_CTRL_RES_NRG_Q + 4 * k);
    }
    in.LTPredCodGain_Q7 = rd_int(ctl, OPUSGPU_REF_OFF_CTRL_LTP_RED_COD_GAIN_Q7); in.signalType = (int8_t)ind[OPUSGPU_REF_OFF_SIGNAL_TYPE];
    in.nb_subfr = rd_int(sCmn, OPUSGPU_REF_OFF_NB_SUBFR); in.subfr_length = rd_int(sCmn, OPUSGPU_REF_OFF_SUBFR_LENGTH);
    in.SNR_dB_Q7 = rd_int(sCmn, OPUSGPU_REF_OFF_SNR_DB_Q7); in.LastGainIndex = (int8_t)shp[OPUSGPU_REF_OFF_SHAPE_LAST_GAIN_INDEX];
    in.condCoding = condCoding; in.input_tilt_Q15 = rd_int(sCmn, OPUSGPU_REF_OFF_INPUT_TILT_Q15);
    in.quantOffsetType = (int8_t)ind[OPUSGPU_REF_OFF_QUANT_OFFSET_TYPE];
    in.nStatesDelayedDecision = rd_int(sCmn, OPUSGPU_REF_OFF_N_STATES_DEL_DEC); in.speech_activity_Q8 = rd_int(sCmn, OPUSGPU_REF_OFF_SPEECH_ACTIVITY_Q8);
    in.input_quality_Q14 = rd_int(ctl, OPUSGPU_REF_OFF_CTRL_INPUT_QUALITY_Q14); in.coding_quality_Q14 = rd_int(ctl, OPUSGPU_REF_OFF_CTRL_CODING_QUALITY_Q14);
    const int rc = run_record(&in, &out, (NoState *)nullptr, [](const opusgpu_process_gains_in *i, NoState *, opusgpu_process_gains_out *o) {
        return opusgpu_silk_process_gains_batch(i, o, 1, nullptr);
    });
    opusgpu_set_last_error(rc);
    if (rc != OPUSGPU_OK) return;
    for (int k = 0; k < in.nb_subfr; k++) {
        wr_int(ctl, OPUSGPU_REF_OFF_CTRL_GAINS_Q16 + 4 * k, out.Gains_Q16[k]);
        wr_int(ctl, OPUSGPU_REF_OFF_CTRL_GAINS_UNQ_Q16 + 4 * k, out.GainsUnq_Q16[k]);
        ind[OPUSGPU_REF_OFF_GAINS_INDICES + k] = (char)out.GainsIndices[k];
    }
    ctl[OPUSGPU_REF_OFF_CTRL_LAST_GAIN_INDEX_PREV] = (char)out.lastGainIndexPrev;
    shp[OPUSGPU_REF_OFF_SHAPE_LAST_GAIN_INDEX] = (char)out.LastGainIndex;
    if (in.signalType == 2) ind[OPUSGPU_REF_OFF_QUANT_OFFSET_TYPE] = (char)out.quantOffsetType;
    wr_int(ctl, OPUSGPU_REF_OFF_CTRL_LAMBDA_Q10, out.Lambda_Q10);
}

extern "C" void opusgpu_silk_prefilter_FIX(void *psEnc, const void *psEncCtrl, int32_t xw_Q3[], const int16_t x[])
{
    if (!psEnc || !psEncCtrl || !xw_Q3 || !x) { opusgpu_set_last_error(OPUSGPU_BAD_ARG); return; }
    static_assert(sizeof(opusgpu_prefilter_state) == OPUSGPU_REF_SIZEOF_SILK_PREFILTER_STATE_FIX, "state record = silk_prefilter_state_FIX");
    char *sCmn = (char *)psEnc + OPUSGPU_REF_OFF_FIX_SCMN, *ind = sCmn + OPUSGPU_REF_OFF_INDICES;
    const char *ctl = (const char *)psEncCtrl;
    opusgpu_prefilter_in *in = (opusgpu_prefilter_in *)calloc(1, sizeof(*in));
    opusgpu_prefilter_out *out = (opusgpu_prefilter_out *)calloc(1, sizeof(*out));
    opusgpu_prefilter_state *st = (opusgpu_prefilter_state *)malloc(sizeof(*st));
    if (!in || !out || !st) { free(in); free(out); free(st); opusgpu_set_last_error(OPUSGPU_ALLOC_FAIL); return; }
    in->nb_subfr = rd_int(sCmn, OPUSGPU_REF_OFF_NB_SUBFR); in->subfr_length = rd_int(sCmn, OPUSGPU_REF_OFF_SUBFR_LENGTH);
    in->signalType = (int8_t)ind[OPUSGPU_REF_OFF_SIGNAL_TYPE]; in->warping_Q16 = rd_int(sCmn, OPUSGPU_REF_OFF_WARPING_Q16);
    in->shapingLPCOrder = rd_int(sCmn, OPUSGPU_REF_OFF_SHAPING_LPC_ORDER); in->coding_quality_Q14 = rd_int(ctl, OPUSGPU_REF_OFF_CTRL_CODING_QUALITY_Q14);
    for (int k = 0; k < 4; k++) {
        in->pitchL[k] = rd_int(ctl, OPUSGPU_REF_OFF_CTRL_PITCHL + 4 * k);
        in->HarmShapeGain_Q14[k] = rd_int(ctl, OPUSGPU_REF_OFF_CTRL_HARM_SHAPE_GAIN_Q14 + 4 * k);
        in->HarmBoost_Q14[k] = rd_int(ctl, OPUSGPU_REF_OFF_CTRL_HARM_BOOST_Q14 + 4 * k);
        in->Tilt_Q14[k] = rd_int(ctl, OPUSGPU_REF_OFF_CTRL_TILT_Q14 + 4 * k);
        in->GainsPre_Q14[k] = rd_int(ctl, OPUSGPU_REF_OFF_CTRL_GAINS_PRE_Q14 + 4 * k);
        in->LF_shp_Q14[k] = rd_int(ctl, OPUSGPU_REF_OFF_CTRL_LF_SHP_Q14 + 4 * k);
    }
    memcpy(in->AR1_Q13, ctl + OPUSGPU_REF_OFF_CTRL_AR1_Q13, sizeof(in->AR1_Q13));
    const long fl = (long)in->nb_subfr * in->subfr_length;
    if (in->nb_subfr < 1 || in->subfr_length < 1 || fl > OPUSGPU_SILK_MAX_FRAME) {
        free(in); free(out); free(st);
        opusgpu_set_last_error(OPUSGPU_BAD_ARG);
        return;
    }
    memcpy(in->x, x, sizeof(int16_t) * (size_t)fl);
    memcpy(st, (char *)psEnc + OPUSGPU_REF_OFF_FIX_SPREFILT, sizeof(*st));
    const int rc = run_record(in, out, st, [](const opusgpu_prefilter_in *i, opusgpu_prefilter_state *s, opusgpu_prefilter_out *o) {
        return opusgpu_silk_prefilter_batch(i, s, o, 1, nullptr);
    });
    opusgpu_set_last_error(rc);
    if (rc == OPUSGPU_OK) {
        memcpy((char *)psEnc + OPUSGPU_REF_OFF_FIX_SPREFILT, st, sizeof(*st));
        memcpy(xw_Q3, out->xw_Q3, sizeof(int32_t) * (size_t)fl);
    }
    free(in); free(out); free(st);
}

extern "C" int opusgpu_silk_VAD_GetSA_Q8_c(void *psEncC, const int16_t pIn[])
{
    if (!psEncC || !pIn) { opusgpu_set_last_error(OPUSGPU_BAD_ARG); return -1; }
    static_assert(sizeof(opusgpu_vad_state) == OPUSGPU_REF_SIZEOF_SILK_VAD_STATE, "state record = silk_VAD_state");
    char *sCmn = (char *)psEncC;
    opusgpu_vad_in in;
    opusgpu_vad_out out;
    opusgpu_vad_state st;
    memset(&in, 0, sizeof(in));
    in.frame_length = rd_int(sCmn, OPUSGPU_REF_OFF_FRAME_LENGTH); in.fs_kHz = rd_int(sCmn, OPUSGPU_REF_OFF_FS_KHZ);
    if (in.frame_length < 8 || in.frame_length > OPUSGPU_SILK_MAX_FRAME) { opusgpu_set_last_error(OPUSGPU_BAD_ARG); return -1; }
    memcpy(in.pIn, pIn, sizeof(int16_t) * (size_t)in.frame_length);
    memcpy(&st, sCmn + OPUSGPU_REF_OFF_SVAD, sizeof(st));
    const int rc = run_record(&in, &out, &st, [](const opusgpu_vad_in *i, opusgpu_vad_state *s, opusgpu_vad_out *o) {
        return opusgpu_silk_vad_batch(i, s, o, 1, nullptr);
    });
    opusgpu_set_last_error(rc);
    if (rc != OPUSGPU_OK) return -1;
    memcpy(sCmn + OPUSGPU_REF_OFF_SVAD, &st, sizeof(st));
    wr_int(sCmn, OPUSGPU_REF_OFF_SPEECH_ACTIVITY_Q8, out.speech_activity_Q8);
    wr_int(sCmn, OPUSGPU_REF_OFF_INPUT_TILT_Q15, out.input_tilt_Q15);
    for (int k = 0; k < 4; k++) wr_int(sCmn, OPUSGPU_REF_OFF_INPUT_QUALITY_BANDS_Q15 + 4 * k, out.input_quality_bands_Q15[k]);
    return 0;
}

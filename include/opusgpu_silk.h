/* opusgpu_silk.h -- C-ABI records of the SILK function-level kernels (BASELINE config #4).
 *
 * The north star names only the two fixed-point inner loops of the SILK encoder:
 *   silk_burg_modified_c   opus-fix/silk/fixed/burg_modified_FIX.c:45-275   (macro silk_burg_modified, silk/SigProc_FIX.h:601)
 *   silk_NSQ_c             opus-fix/silk/NSQ.c:74-180 (+ silk_noise_shape_quantizer :183-421, silk_nsq_scale_states :423-496)
 * Their inputs are produced by SILK analysis code that is out of scope, so the batched
 * entry points take "function-boundary records": flat, pointer-free copies of the arguments of one call,
 * exactly what the reference passes at silk/fixed/find_LPC_FIX.c:63,69 and silk/fixed/encode_frame_FIX.c:317.
 */
#ifndef OPUSGPU_SILK_H
#define OPUSGPU_SILK_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define OPUSGPU_SILK_MAX_ORDER 16
#define OPUSGPU_SILK_BURG_MAX_X 384          /* MAX_FRAME_SIZE, burg_modified_FIX.c:36 */
#define OPUSGPU_SILK_MAX_FRAME 320           /* MAX_FRAME_LENGTH = 20 ms @ 16 kHz (silk/define.h:92) */

/* one silk_burg_modified() call */
typedef struct opusgpu_burg_in {
    int16_t x[OPUSGPU_SILK_BURG_MAX_X];      /* nb_subfr * subfr_length samples */
    int32_t minInvGain_Q30;
    int32_t subfr_length;                    /* incl. D preceding samples */
    int32_t nb_subfr;
    int32_t D;                               /* order (<= 16) */
} opusgpu_burg_in;

typedef struct opusgpu_burg_out {
    int32_t res_nrg;
    int32_t res_nrg_Q;
    int32_t A_Q16[OPUSGPU_SILK_MAX_ORDER];
} opusgpu_burg_out;

/* silk_nsq_state (opus-fix/silk/structs.h), identical layout */
typedef struct opusgpu_nsq_state {
    int16_t xq[2 * OPUSGPU_SILK_MAX_FRAME];
    int32_t sLTP_shp_Q14[2 * OPUSGPU_SILK_MAX_FRAME];
    int32_t sLPC_Q14[80 + 32];               /* MAX_SUB_FRAME_LENGTH + NSQ_LPC_BUF_LENGTH */
    int32_t sAR2_Q14[16];
    int32_t sLF_AR_shp_Q14;
    int32_t lagPrev;
    int32_t sLTP_buf_idx;
    int32_t sLTP_shp_buf_idx;
    int32_t rand_seed;
    int32_t prev_gain_Q16;
    int32_t rewhite_flag;
} opusgpu_nsq_state;

/* one silk_NSQ() call: the fields of psEncC / psIndices it reads, then its array arguments */
typedef struct opusgpu_nsq_in {
    int32_t nb_subfr, subfr_length, frame_length, ltp_mem_length, predictLPCOrder, shapingLPCOrder;
    int32_t signalType, quantOffsetType, NLSFInterpCoef_Q2, Seed;
    int32_t Lambda_Q10, LTP_scale_Q14;
    int32_t HarmShapeGain_Q14[4], Tilt_Q14[4], LF_shp_Q14[4], Gains_Q16[4], pitchL[4];
    int32_t x_Q3[OPUSGPU_SILK_MAX_FRAME];
    int16_t PredCoef_Q12[2 * 16];
    int16_t LTPCoef_Q14[5 * 4];
    int16_t AR2_Q13[4 * 16];
} opusgpu_nsq_in;

typedef struct opusgpu_nsq_out {
    int8_t pulses[OPUSGPU_SILK_MAX_FRAME];
} opusgpu_nsq_out;

/* Batched entry points (device pointers, asynchronous on hip_stream). One record per call of the
 * reference function; NSQ states are updated in place, as the reference updates *NSQ.
 * d_workspace: device scratch of opusgpu_silk_nsq_workspace_bytes(n) bytes (the re-whitening buffers
 * sLTP / sLTP_Q15 that the reference allocates on its stack, silk/NSQ.c:117-120). */
int opusgpu_silk_burg_modified_batch(const opusgpu_burg_in *d_in, opusgpu_burg_out *d_out, int n, void *hip_stream);
size_t opusgpu_silk_nsq_workspace_bytes(int n);
int opusgpu_silk_nsq_batch(const opusgpu_nsq_in *d_in, opusgpu_nsq_state *d_state, opusgpu_nsq_out *d_out, int n,
                           void *d_workspace, size_t workspace_bytes, void *hip_stream);

/* The header fields of the records arrive in device memory, so they are bounds-checked on the device, per record
 * (concentus_amd/csrc/silk_validate.h: nb_subfr <= 4, subfr_length <= 80, orders <= 16, pitch lags inside the LTP
 * memory, 1..4 delayed-decision states ...). A record that fails is SKIPPED -- Burg: A_Q16 = 0, res_nrg = 0,
 * res_nrg_Q = INT32_MIN; NSQ / NSQ_del_dec: pulses zeroed, state left untouched -- and counted.
 * opusgpu_silk_bad_records() waits for hip_stream and returns the number of records skipped on the current device
 * since the previous call (>= 0), or a negative OPUSGPU_* code; it resets the counter. */
int opusgpu_silk_bad_records(void *hip_stream);

/* Per-call hook with the reference's own signature (host pointers, one call = one record, synchronous): what the macro
 * silk_burg_modified (opus-fix/silk/SigProc_FIX.h:601-602) would be pointed at. Plumbing / parity only. Errors are
 * reported through opusgpu_get_last_error(). */
void opusgpu_silk_burg_modified_c(int32_t *res_nrg, int *res_nrg_Q, int32_t A_Q16[], const int16_t x[],
                                  const int32_t minInvGain_Q30, const int subfr_length, const int nb_subfr,
                                  const int D, int arch);

/* ---- silk_NSQ_del_dec(): the quantizer the reference uses at complexity >= 4 (silk/fixed/encode_frame_FIX.c:311,
 * silk/control_codec.c:345-377: 2, 3 or 4 delayed-decision states, warped noise shaping) ----
 *   silk_NSQ_del_dec_c                    opus-fix/silk/NSQ_del_dec.c:112-318 (macro silk_NSQ_del_dec, silk/main.h:271-296)
 *   silk_noise_shape_quantizer_del_dec    :324-630,   silk_nsq_del_dec_scale_states :632-724
 * One record = the arguments of one call: everything silk_NSQ() takes plus the two psEncC fields only this
 * quantizer reads. The NSQ state record is the same opusgpu_nsq_state. */
#define OPUSGPU_SILK_MAX_DEL_DEC_STATES 4    /* MAX_DEL_DEC_STATES, silk/define.h:160 */
typedef struct opusgpu_nsq_dd_in {
    opusgpu_nsq_in base;
    int32_t nStatesDelayedDecision;          /* 1..4 */
    int32_t warping_Q16;
} opusgpu_nsq_dd_in;

typedef struct opusgpu_nsq_dd_out {
    int8_t pulses[OPUSGPU_SILK_MAX_FRAME];
    int32_t Seed;                            /* psIndices->Seed after the call (the winner's initial seed, NSQ_del_dec.c:297) */
} opusgpu_nsq_dd_out;

size_t opusgpu_silk_nsq_del_dec_workspace_bytes(int n);
int opusgpu_silk_nsq_del_dec_batch(const opusgpu_nsq_dd_in *d_in, opusgpu_nsq_state *d_state, opusgpu_nsq_dd_out *d_out, int n,
                                   void *d_workspace, size_t workspace_bytes, void *hip_stream);

/* ---- silk_find_LPC_FIX(): the analysis step around silk_burg_modified (first slice) ----------------
 *   silk_find_LPC_FIX   opus-fix/silk/fixed/find_LPC_FIX.c:37-151, called at silk/fixed/find_pred_coefs_FIX.c:136
 * Burg analysis of the whole frame; with NLSF interpolation enabled (complexity >= 4, control_codec.c:345-377) a second Burg
 * analysis of the last 10 ms, silk_A2NLSF, and the search over the four interpolation factors (silk_interpolate ->
 * silk_NLSF2A -> silk_LPC_analysis_filter -> silk_sum_sqr_shift per factor); finally silk_A2NLSF of the winner. One record =
 * the arguments of one call plus the psEncC fields it reads; the output is what the call leaves in NLSF_Q15[] and in
 * psEncC->indices.NLSFInterpCoef_Q2. Records whose header would index out of bounds are skipped (status OPUSGPU_BAD_ARG,
 * outputs zero) and counted by opusgpu_silk_bad_records(). */
typedef struct opusgpu_find_lpc_in {
    int16_t x[OPUSGPU_SILK_BURG_MAX_X];      /* LPC_in_pre: nb_subfr * (subfr_length + predictLPCOrder) samples */
    int32_t minInvGain_Q30;
    int32_t subfr_length;                    /* psEncC->subfr_length (WITHOUT the order samples: 80 at 16 kHz) */
    int32_t nb_subfr;                        /* 2 or 4 */
    int32_t predictLPCOrder;                 /* 10 or 16 */
    int32_t useInterpolatedNLSFs;
    int32_t first_frame_after_reset;
    int16_t prev_NLSFq_Q15[OPUSGPU_SILK_MAX_ORDER];
    int32_t reserved[2];
} opusgpu_find_lpc_in;

typedef struct opusgpu_find_lpc_out {
    int16_t NLSF_Q15[OPUSGPU_SILK_MAX_ORDER];
    int32_t NLSFInterpCoef_Q2;               /* 0..4 (4 = no interpolation) */
    int32_t status;                          /* OPUSGPU_OK, or OPUSGPU_BAD_ARG for a skipped record */
} opusgpu_find_lpc_out;

int opusgpu_silk_find_lpc_batch(const opusgpu_find_lpc_in *d_in, opusgpu_find_lpc_out *d_out, int n, void *hip_stream);

/* ---- silk_process_NLSFs, batched (second slice) -------------------------------------------------------
 * Replaces silk_process_NLSFs(psEncC, PredCoef_Q12, pNLSF_Q15, prev_NLSFq_Q15) (opus-fix/silk/process_NLSFs.c:35-106, called
 * at silk/fixed/find_pred_coefs_FIX.c:139): NLSF weights, first-stage VQ, survivor trellis (silk_NLSF_encode /
 * silk_NLSF_del_dec_quant), decode, silk_NLSF2A for both frame halves. A record is the arguments of one call plus the
 * psEncC fields it reads; the codebook follows predictLPCOrder (16: silk_NLSF_CB_WB, 10: silk_NLSF_CB_NB_MB). */
typedef struct opusgpu_process_nlsf_in {
    int16_t NLSF_Q15[OPUSGPU_SILK_MAX_ORDER];        /* pNLSF_Q15 on entry (from silk_find_LPC_FIX) */
    int16_t prev_NLSFq_Q15[OPUSGPU_SILK_MAX_ORDER];
    int32_t speech_activity_Q8;                      /* 0..256 */
    int32_t nb_subfr;                                /* 2 or 4 */
    int32_t predictLPCOrder;                         /* 10 or 16 */
    int32_t useInterpolatedNLSFs;
    int32_t NLSFInterpCoef_Q2;                       /* psEncC->indices.NLSFInterpCoef_Q2, 0..4 */
    int32_t NLSF_MSVQ_Survivors;                     /* 1..32 */
    int32_t signalType;                              /* 0..2 */
    int32_t reserved;
} opusgpu_process_nlsf_in;

typedef struct opusgpu_process_nlsf_out {
    int16_t PredCoef_Q12[2][OPUSGPU_SILK_MAX_ORDER];
    int16_t NLSF_Q15[OPUSGPU_SILK_MAX_ORDER];        /* the quantised NLSFs (pNLSF_Q15 on return) */
    int8_t NLSFIndices[OPUSGPU_SILK_MAX_ORDER + 1];  /* psEncC->indices.NLSFIndices */
    int8_t pad[3];
    int32_t status;
} opusgpu_process_nlsf_out;

int opusgpu_silk_process_nlsfs_batch(const opusgpu_process_nlsf_in *d_in, opusgpu_process_nlsf_out *d_out, int n, void *hip_stream);

/* ---- silk_residual_energy_FIX, batched ---------------------------------------------------------------------------------
 * Replaces silk_residual_energy_FIX(nrgs, nrgsQ, x, a_Q12, gains, subfr_length, nb_subfr, LPC_order, arch)
 * (opus-fix/silk/fixed/residual_energy_FIX.c:37-98, called at silk/fixed/find_pred_coefs_FIX.c:142). */
typedef struct opusgpu_res_nrg_in {
    int16_t x[OPUSGPU_SILK_BURG_MAX_X];              /* LPC_in_pre: nb_subfr * (subfr_length + LPC_order) samples */
    int16_t a_Q12[2][OPUSGPU_SILK_MAX_ORDER];
    int32_t gains[4];
    int32_t subfr_length;
    int32_t nb_subfr;                                /* 2 or 4 */
    int32_t LPC_order;
    int32_t reserved;
} opusgpu_res_nrg_in;

typedef struct opusgpu_res_nrg_out {
    int32_t nrgs[4];
    int32_t nrgsQ[4];
    int32_t status;
    int32_t reserved;
} opusgpu_res_nrg_out;

int opusgpu_silk_residual_energy_batch(const opusgpu_res_nrg_in *d_in, opusgpu_res_nrg_out *d_out, int n, void *hip_stream);

/* ---- silk_find_pred_coefs_FIX, batched (third slice) ----------------------------------------------------
 * Replaces silk_find_pred_coefs_FIX(psEnc, psEncCtrl, res_pitch, x, condCoding) (opus-fix/silk/fixed/find_pred_coefs_FIX.c:35-148,
 * called at silk/fixed/encode_frame_FIX.c) WHOLE: the voiced branch (silk_find_LTP_FIX, silk_quant_LTP_gains,
 * silk_LTP_scale_ctrl_FIX, silk_LTP_analysis_filter_FIX) or the unvoiced one, then silk_find_LPC_FIX, silk_process_NLSFs and
 * silk_residual_energy_FIX. A record is the arguments of one call plus the psEnc / psEncCtrl fields it reads; the output is
 * every field it writes. For an unvoiced frame the call does not touch LTP_scale_Q14 / indices.LTP_scaleIndex: reported as
 * 0 / -1. */
#define OPUSGPU_SILK_MAX_LTP_MEM 320             /* LTP_MEM_LENGTH_MS * MAX_FS_KHZ */
typedef struct opusgpu_find_pred_coefs_in {
    int16_t res_pitch[OPUSGPU_SILK_MAX_LTP_MEM + OPUSGPU_SILK_MAX_FRAME]; /* res_pitch[0 .. ltp_mem_length + frame_length) */
    int16_t x[OPUSGPU_SILK_MAX_LTP_MEM + OPUSGPU_SILK_MAX_FRAME];         /* x[-ltp_mem_length .. frame_length): x_buf up to the end of the frame */
    int32_t Gains_Q16[4];                    /* psEncCtrl->Gains_Q16 */
    int32_t pitchL[4];                       /* psEncCtrl->pitchL */
    int16_t prev_NLSFq_Q15[OPUSGPU_SILK_MAX_ORDER];
    int32_t nb_subfr, subfr_length, predictLPCOrder, ltp_mem_length;
    int32_t signalType, condCoding, first_frame_after_reset, useInterpolatedNLSFs;
    int32_t speech_activity_Q8, NLSF_MSVQ_Survivors, mu_LTP_Q9, LTPQuantLowComplexity;
    int32_t sum_log_gain_Q7, coding_quality_Q14, PacketLoss_perc, nFramesPerPacket;
} opusgpu_find_pred_coefs_in;

typedef struct opusgpu_find_pred_coefs_out {
    int16_t PredCoef_Q12[2][OPUSGPU_SILK_MAX_ORDER];
    int16_t LTPCoef_Q14[20];
    int16_t NLSF_Q15[OPUSGPU_SILK_MAX_ORDER];    /* -> psEnc->sCmn.prev_NLSFq_Q15 */
    int32_t ResNrg[4];
    int32_t ResNrgQ[4];
    int32_t LTPredCodGain_Q7, LTP_scale_Q14, sum_log_gain_Q7;
    int8_t NLSFIndices[OPUSGPU_SILK_MAX_ORDER + 1];
    int8_t NLSFInterpCoef_Q2;
    int8_t LTPIndex[4];
    int8_t PERIndex;
    int8_t LTP_scaleIndex;
    int32_t status;
} opusgpu_find_pred_coefs_out;

int opusgpu_silk_find_pred_coefs_batch(const opusgpu_find_pred_coefs_in *d_in, opusgpu_find_pred_coefs_out *d_out, int n, void *hip_stream);

/* ---- silk_process_gains_FIX, batched (fourth slice) -----------------------------------------------------
 * Replaces silk_process_gains_FIX(psEnc, psEncCtrl, condCoding) (opus-fix/silk/fixed/process_gains_FIX.c:37-125, called from
 * silk_encode_frame_FIX between silk_find_pred_coefs_FIX and the noise-shaping quantiser): LTP-gain dependent gain
 * reduction, the soft limit against the residual energies, silk_gains_quant, quantOffsetType for voiced frames, Lambda_Q10. */
typedef struct opusgpu_process_gains_in {
    int32_t Gains_Q16[4];                    /* psEncCtrl->Gains_Q16 */
    int32_t ResNrg[4];                       /* psEncCtrl->ResNrg (from silk_find_pred_coefs_FIX) */
    int32_t ResNrgQ[4];
    int32_t LTPredCodGain_Q7, signalType, nb_subfr, subfr_length;
    int32_t SNR_dB_Q7, LastGainIndex /* psEnc->sShape.LastGainIndex */, condCoding, input_tilt_Q15;
    int32_t quantOffsetType, nStatesDelayedDecision, speech_activity_Q8, input_quality_Q14;
    int32_t coding_quality_Q14, reserved[3];
} opusgpu_process_gains_in;

typedef struct opusgpu_process_gains_out {
    int32_t Gains_Q16[4];                    /* quantised */
    int32_t GainsUnq_Q16[4];
    int32_t Lambda_Q10, LastGainIndex, lastGainIndexPrev, quantOffsetType;
    int8_t GainsIndices[4];
    int32_t status;
} opusgpu_process_gains_out;

int opusgpu_silk_process_gains_batch(const opusgpu_process_gains_in *d_in, opusgpu_process_gains_out *d_out, int n, void *hip_stream);

/* ---- silk_noise_shape_analysis_FIX, batched (fifth slice) -----------------------------------------------
 * Replaces silk_noise_shape_analysis_FIX(psEnc, psEncCtrl, pitch_res, x, arch) (opus-fix/silk/fixed/noise_shape_analysis_FIX.c:146-466,
 * called from silk_encode_frame_FIX before silk_find_pred_coefs_FIX): windowing, (warped) autocorrelation, Schur recursion, the
 * two bandwidth-expanded shaping filters per subframe, the initial gains, low-frequency / tilt / harmonic shaping controls and
 * their smoothing state. A record is the two signal arguments plus the psEnc / psEncCtrl fields read; the output is every
 * field written. AR1_Q13 / AR2_Q13 rows hold shapingLPCOrder entries each (stride 16, the rest zero). */
#define OPUSGPU_SILK_MAX_LA_SHAPE 80             /* LA_SHAPE_MS * MAX_FS_KHZ */
typedef struct opusgpu_noise_shape_in {
    int16_t x[OPUSGPU_SILK_MAX_FRAME + 2 * OPUSGPU_SILK_MAX_LA_SHAPE]; /* x[-la_shape .. frame_length + la_shape) */
    int16_t pitch_res[OPUSGPU_SILK_MAX_FRAME];                          /* the frame's pitch-analysis residual */
    int32_t fs_kHz, nb_subfr, subfr_length, la_shape;
    int32_t shapeWinLength, shapingLPCOrder, warping_Q16, SNR_dB_Q7;
    int32_t useCBR, speech_activity_Q8, signalType, LTPCorr_Q15;       /* LTPCorr_Q15: psEnc->LTPCorr_Q15 */
    int32_t input_quality_bands_Q15[2], predGain_Q16, reserved;        /* predGain_Q16: psEncCtrl->predGain_Q16 */
    int32_t pitchL[4];
    int32_t HarmBoost_smth_Q16, HarmShapeGain_smth_Q16, Tilt_smth_Q16, reserved2;   /* psEnc->sShape */
} opusgpu_noise_shape_in;

typedef struct opusgpu_noise_shape_out {
    int32_t Gains_Q16[4];
    int32_t GainsPre_Q14[4];
    int16_t AR1_Q13[4 * 16];
    int16_t AR2_Q13[4 * 16];
    int32_t LF_shp_Q14[4];
    int32_t HarmBoost_Q14[4];
    int32_t HarmShapeGain_Q14[4];
    int32_t Tilt_Q14[4];
    int32_t HarmBoost_smth_Q16, HarmShapeGain_smth_Q16, Tilt_smth_Q16;
    int32_t input_quality_Q14, coding_quality_Q14, sparseness_Q8, quantOffsetType;
    int32_t status;
} opusgpu_noise_shape_out;

int opusgpu_silk_noise_shape_analysis_batch(const opusgpu_noise_shape_in *d_in, opusgpu_noise_shape_out *d_out, int n, void *hip_stream);

/* ---- silk_prefilter_FIX, batched (sixth slice) ----------------------------------------------------------
 * Replaces silk_prefilter_FIX(psEnc, psEncCtrl, xw_Q3, x) (opus-fix/silk/fixed/prefilter_FIX.c:102-184, called from
 * silk_encode_frame_FIX right before the noise-shaping quantiser): warped short-term analysis filter, harmonic high-pass,
 * tilt / low-frequency / harmonic shaping. The state record has the layout of silk_prefilter_state_FIX (structs_FIX.h:53-62)
 * and is updated in place; the output is xw_Q3[frame_length], the x_Q3 argument of silk_NSQ / silk_NSQ_del_dec.
 * Records whose pitch lags (pitchL[], lagPrev) are 1 or above 316 are rejected (status OPUSGPU_BAD_ARG, state untouched): the
 * kernel keeps the newest 320 entries of the harmonic-shaping ring per frame; the encoder's lags are 0 or 32 .. 288. */
typedef struct opusgpu_prefilter_state {
    int16_t sLTP_shp[512];
    int32_t sAR_shp[17];
    int32_t sLTP_shp_buf_idx;
    int32_t sLF_AR_shp_Q12, sLF_MA_shp_Q12, sHarmHP_Q2, rand_seed;
    int32_t lagPrev;
} opusgpu_prefilter_state;

typedef struct opusgpu_prefilter_in {
    int16_t x[OPUSGPU_SILK_MAX_FRAME];
    int16_t AR1_Q13[4 * 16];                 /* psEncCtrl->AR1_Q13 */
    int32_t pitchL[4], HarmShapeGain_Q14[4], HarmBoost_Q14[4], Tilt_Q14[4], GainsPre_Q14[4], LF_shp_Q14[4];
    int32_t coding_quality_Q14, nb_subfr, subfr_length, signalType;
    int32_t warping_Q16, shapingLPCOrder, reserved[2];
} opusgpu_prefilter_in;

typedef struct opusgpu_prefilter_out {
    int32_t xw_Q3[OPUSGPU_SILK_MAX_FRAME];
    int32_t status;
    int32_t reserved[3];
} opusgpu_prefilter_out;

int opusgpu_silk_prefilter_batch(const opusgpu_prefilter_in *d_in, opusgpu_prefilter_state *d_state, opusgpu_prefilter_out *d_out, int n,
                                 void *hip_stream);

/* ---- silk_find_pitch_lags_FIX, batched (seventh slice) --------------------------------------------------
 * Replaces silk_find_pitch_lags_FIX(psEnc, psEncCtrl, res, x, arch) (opus-fix/silk/fixed/find_pitch_lags_FIX.c:37-145, the first
 * analysis call of silk_encode_frame_FIX) including silk_pitch_analysis_core (silk/fixed/pitch_analysis_core_FIX.c:86-581):
 * windowed autocorrelation, Schur recursion, LPC whitening of the whole pitch buffer (res[], which the later analysis calls
 * read), the three-stage pitch search at 4 / 8 / input kHz and the voicing decision. 8 and 16 kHz (fs_kHz 12 -> BAD_ARG). */
#define OPUSGPU_SILK_MAX_LA_PITCH 32             /* LA_PITCH_MS * MAX_FS_KHZ */
#define OPUSGPU_SILK_PITCH_BUF (OPUSGPU_SILK_MAX_LA_PITCH + OPUSGPU_SILK_MAX_FRAME + OPUSGPU_SILK_MAX_LTP_MEM)
typedef struct opusgpu_find_pitch_lags_in {
    int16_t x_buf[OPUSGPU_SILK_PITCH_BUF];   /* x[-ltp_mem_length .. frame_length + la_pitch) */
    int32_t fs_kHz, nb_subfr, frame_length, ltp_mem_length;
    int32_t la_pitch, pitch_LPC_win_length, pitchEstimationLPCOrder, pitchEstimationComplexity;
    int32_t pitchEstimationThreshold_Q16, signalType, first_frame_after_reset, speech_activity_Q8;
    int32_t prevSignalType, input_tilt_Q15, prevLag, LTPCorr_Q15;   /* LTPCorr_Q15: psEnc->LTPCorr_Q15 of the previous frame */
} opusgpu_find_pitch_lags_in;

typedef struct opusgpu_find_pitch_lags_out {
    int16_t res[OPUSGPU_SILK_PITCH_BUF];     /* the whitened buffer (res_pitch) */
    int32_t pitchL[4];
    int32_t lagIndex, contourIndex, LTPCorr_Q15, signalType;
    int32_t predGain_Q16, status, reserved[2];
} opusgpu_find_pitch_lags_out;

int opusgpu_silk_find_pitch_lags_batch(const opusgpu_find_pitch_lags_in *d_in, opusgpu_find_pitch_lags_out *d_out, int n, void *hip_stream);

/* ---- silk_encode_indices + silk_encode_pulses, batched (eighth slice) -------------------------------------
 * Replace silk_encode_indices(psEncC, psRangeEnc, FrameIndex, 0, condCoding) (opus-fix/silk/encode_indices.c:36-193) and
 * silk_encode_pulses(psRangeEnc, signalType, quantOffsetType, pulses, frame_length) (silk/encode_pulses.c:64-206, with the shell
 * and sign coders), the two calls silk_encode_frame_FIX makes after the quantiser (silk/fixed/encode_frame_FIX.c:328-336): the
 * side information and the excitation of one frame onto the Opus range coder. The range coder travels as a record: the fields
 * of the reference's ec_ctx (celt/entcode.h:63-94) and its buffer (head bytes [0, offs), tail bytes [storage - end_offs,
 * storage)), updated in place. `which`: 1 = indices only, 2 = pulses only, 3 = both, in that order. frame_length a multiple of
 * 16 (8 / 16 kHz). */
#define OPUSGPU_EC_BUF 1280
typedef struct opusgpu_ec_state {
    uint32_t storage, end_offs, end_window;
    int32_t nend_bits, nbits_total;
    uint32_t offs, rng, val, ext;
    int32_t rem, error;
    uint32_t reserved;
    uint8_t buf[OPUSGPU_EC_BUF];             /* storage <= OPUSGPU_EC_BUF */
} opusgpu_ec_state;

typedef struct opusgpu_silk_bits_in {
    int8_t pulses[OPUSGPU_SILK_MAX_FRAME];
    int8_t GainsIndices[4], LTPIndex[4], NLSFIndices[OPUSGPU_SILK_MAX_ORDER + 1], pad[3];
    int32_t lagIndex, contourIndex, signalType, quantOffsetType;
    int32_t NLSFInterpCoef_Q2, PERIndex, LTP_scaleIndex, Seed;
    int32_t nb_subfr, fs_kHz, predictLPCOrder, frame_length;
    int32_t condCoding, ec_prevSignalType, ec_prevLagIndex, which;
    int32_t reserved;
} opusgpu_silk_bits_in;

typedef struct opusgpu_silk_bits_out {
    int32_t ec_prevSignalType, ec_prevLagIndex;   /* psEncC->ec_prevSignalType / ec_prevLagIndex after silk_encode_indices */
    int32_t status, reserved;
} opusgpu_silk_bits_out;

int opusgpu_silk_encode_bits_batch(const opusgpu_silk_bits_in *d_in, opusgpu_ec_state *d_ec, opusgpu_silk_bits_out *d_out, int n,
                                   void *hip_stream);

/* ---- silk_VAD_GetSA_Q8, batched (ninth slice) -----------------------------------------------------------
 * Replaces silk_VAD_GetSA_Q8_c(psEncC, pIn) (opus-fix/silk/VAD.c:82-312; macro silk_VAD_GetSA_Q8, silk/main.h:305-312, called at
 * silk/fixed/encode_frame_FIX.c:58): three-stage analysis filter bank, band energies, noise-level tracking, speech activity,
 * spectral tilt and the per-band input quality -- what the analysis chain reads as speech_activity_Q8 / input_tilt_Q15 /
 * input_quality_bands_Q15. The state record has the layout of silk_VAD_state (silk/structs.h:60-73) and is updated in place. */
typedef struct opusgpu_vad_state {
    int32_t AnaState[2], AnaState1[2], AnaState2[2], XnrgSubfr[4], NrgRatioSmth_Q8[4];
    int16_t HPstate, pad;
    int32_t NL[4], inv_NL[4], NoiseLevelBias[4], counter;
} opusgpu_vad_state;

typedef struct opusgpu_vad_in {
    int16_t pIn[OPUSGPU_SILK_MAX_FRAME];     /* psEncC->inputBuf + 1: frame_length samples */
    int32_t frame_length, fs_kHz, reserved[2];
} opusgpu_vad_in;

typedef struct opusgpu_vad_out {
    int32_t speech_activity_Q8, input_tilt_Q15;
    int32_t input_quality_bands_Q15[4];
    int32_t status, reserved;
} opusgpu_vad_out;

int opusgpu_silk_vad_batch(const opusgpu_vad_in *d_in, opusgpu_vad_state *d_state, opusgpu_vad_out *d_out, int n, void *hip_stream);

/* ---- the bitrate-control loop of silk_encode_frame_FIX, batched (tenth slice) ---------------------------
 * Replaces the control arithmetic of the `for( iter = 0; ; iter++ )` loop of silk_encode_frame_FIX
 * (opus-fix/silk/fixed/encode_frame_FIX.c:276-423): after every quantise + entropy-code pass of a frame it compares the
 * coder's ec_tell() with maxBits, keeps the lower / upper bracket, moves the gain multiplier (rate/distortion slope or
 * interpolation), raises Lambda_Q10, re-quantises the gains (silk_gains_quant, silk/gain_quant.c:41) and says what the host
 * must do before the next pass. One record per frame carries the loop's locals between passes; zero `started` before the
 * first call. A step consumes the pass just coded (the frame's opusgpu_ec_state) and runs on until the frame is finished
 * (`done`) or needs another pass (`recode`: quantise with Gains_Q16 / Lambda_Q10, code with GainsIndices, both from the
 * coder / quantiser state the frame ENTERED with); `save2` asks to keep a copy of the coder + quantiser state of the pass
 * just consumed (the loop's sRangeEnc_copy2 / sNSQ_copy2, :389-395), `restore2` to make that copy the frame's result (:361-368).
 * Finished frames are left untouched by later steps. */
typedef struct opusgpu_silk_rate_ctl {
    int32_t maxBits, useCBR, condCoding, nb_subfr;       /* arguments of silk_encode_frame_FIX / psEnc->sCmn.nb_subfr */
    int32_t frame_length, started, reserved0[2];
    int32_t GainsUnq_Q16[4];                             /* sEncCtrl.GainsUnq_Q16 (silk_process_gains_FIX) */
    int32_t Gains_Q16[4];                                /* out: sEncCtrl.Gains_Q16 of the next pass */
    int32_t lastGainIndexPrev, LastGainIndex;            /* sEncCtrl.lastGainIndexPrev; psEnc->sShape.LastGainIndex (in / out) */
    int32_t Lambda_Q10;                                  /* in / out */
    int8_t GainsIndices[4];                              /* psEnc->sCmn.indices.GainsIndices (in / out) */
    int32_t iter, gainMult_Q8, found_lower, found_upper; /* the loop's locals */
    int32_t nBits_lower, nBits_upper, gainMult_lower, gainMult_upper;
    int32_t gainsID, gainsID_lower, gainsID_upper, LastGainIndex_copy2;
    int32_t done, recode, save2, restore2;               /* out: decisions of this step */
    int32_t nBits, passes, status, reserved1;            /* ec_tell() of the pass consumed; passes coded so far */
} opusgpu_silk_rate_ctl;

int opusgpu_silk_rate_control_batch(opusgpu_silk_rate_ctl *d_ctl, const opusgpu_ec_state *d_ec, int n, void *hip_stream);

/* ---- the first pass of silk_encode_frame_FIX for a batch of frames, one call ------------------------------
 * silk_find_pitch_lags_FIX -> silk_noise_shape_analysis_FIX -> silk_find_pred_coefs_FIX -> silk_process_gains_FIX ->
 * silk_prefilter_FIX -> silk_NSQ / silk_NSQ_del_dec -> silk_encode_indices + silk_encode_pulses
 * (opus-fix/silk/fixed/encode_frame_FIX.c:176-336) over n frames on one stream: the batched entry points above back to back, each
 * record completed on the device from the outputs of the earlier stages (the fields the reference passes on through psEnc /
 * psEncCtrl). The caller fills what a frame brings with it -- the pitch buffer, x, the configuration and VAD fields, the states the
 * previous frame left -- in every *_in record; the fields the chain fills may be left zero (they are listed per record in
 * concentus_amd/silk_chain.py: CHAIN_FED_FIELDS). All pointers are device pointers to n records; every *_out buffer is written;
 * prefilter_state / nsq_state / ec_state are updated in place. q_in / q_out: opusgpu_nsq_in / opusgpu_nsq_out records, or
 * opusgpu_nsq_dd_in / opusgpu_nsq_dd_out when del_dec. bits_in NULL: stop after the quantiser. workspace: the quantiser's scratch
 * (opusgpu_silk_nsq_workspace_bytes / opusgpu_silk_nsq_del_dec_workspace_bytes). The bitrate loop sits on top of this call:
 * opusgpu_silk_rate_control_batch after it, then quantiser + coder again on the frames that ask for it. */
typedef struct opusgpu_silk_chain_bufs {
    const opusgpu_find_pitch_lags_in *pitch_in; opusgpu_find_pitch_lags_out *pitch_out;
    opusgpu_noise_shape_in *shape_in; opusgpu_noise_shape_out *shape_out;
    opusgpu_find_pred_coefs_in *fpc_in; opusgpu_find_pred_coefs_out *fpc_out;
    opusgpu_process_gains_in *gains_in; opusgpu_process_gains_out *gains_out;
    opusgpu_prefilter_in *prefilter_in; opusgpu_prefilter_state *prefilter_state; opusgpu_prefilter_out *prefilter_out;
    void *q_in; opusgpu_nsq_state *nsq_state; void *q_out;
    opusgpu_silk_bits_in *bits_in; opusgpu_ec_state *ec_state; opusgpu_silk_bits_out *bits_out;
    void *workspace; size_t workspace_bytes;
} opusgpu_silk_chain_bufs;

int opusgpu_silk_encode_frames_batch(const opusgpu_silk_chain_bufs *bufs, int fs_kHz, int nb_subfr, int del_dec, int n, void *hip_stream);

/* ---- silk_encode_frame_FIX WITH its bitrate loop for a batch of frames, one call (encode_frame_FIX.c:176-423) --------------------
 * opusgpu_silk_encode_frames_batch, then the reference's `for( iter = 0; ; iter++ )` loop (:276-423) on the device: a rate-control
 * step over the batch after every pass (opusgpu_silk_rate_control_batch); its decisions are carried out by a kernel -- keep / restore
 * the lower-bracket copy of coder + quantiser state (:389-395, :361-368), put the frames that go again back to the state they
 * entered with (:283-289) and on a list -- and quantiser + coder run again over that list IN PLACE (index indirection, no gathered
 * copies). The host reads one number per iteration (the length of the list) and stops when it is zero.
 * bufs as above with bits_in / ec_state / bits_out set; d_ctl: n records with maxBits, useCBR, condCoding, nb_subfr, frame_length set
 * and the rest zero -- they return LastGainIndex, GainsIndices, Lambda_Q10 and `passes` per frame; nsq_state / ec_state / q_out
 * (pulses, Seed) / bits_in end as silk_encode_frame_FIX leaves them. d_loop_workspace: opusgpu_silk_encode_frames_cbr_workspace_bytes(n)
 * bytes of device memory (the per-frame copies the reference keeps on its stack). *passes (may be NULL): iterations that re-coded
 * at least one frame. */
size_t opusgpu_silk_encode_frames_cbr_workspace_bytes(int n);
int opusgpu_silk_encode_frames_cbr_batch(const opusgpu_silk_chain_bufs *bufs, opusgpu_silk_rate_ctl *d_ctl, int fs_kHz, int nb_subfr,
                                         int del_dec, int n, void *d_loop_workspace, size_t loop_workspace_bytes, int *passes,
                                         void *hip_stream);

/* ---- streams mode: the state silk_encode_frame_FIX carries from frame to frame, on the device -----------------
 * One record per stream holds what the frame function reads from the previous frame through psEnc (opus-fix/silk/fixed/
 * encode_frame_FIX.c:128, :145, :427, :437-441 and the fields its analysis calls update): the tail of x_buf, prevLag / prevSignalType /
 * first_frame_after_reset, LTPCorr_Q15, the noise-shaping smoothers and LastGainIndex of sShape, prev_NLSFq_Q15, sum_log_gain_Q7, the
 * entropy coder's conditional-coding memory and frameCounter. The two big states (silk_prefilter_state_FIX, silk_nsq_state) are the
 * opusgpu_prefilter_state / opusgpu_nsq_state arrays the chain already updates in place: keep one per stream.
 * Per frame:  opusgpu_silk_stream_carry_in   (stream record + the frame's samples d_input[n][OPUSGPU_SILK_MAX_FRAME] -> the carried
 *                                              fields of the frame's *_in records: x_buf and its slices, the scalars above, Seed)
 *             opusgpu_silk_encode_frames_batch or ..._cbr_batch
 *             opusgpu_silk_stream_carry_out  (the frame's outputs -> stream record; d_ctl = the bitrate loop's records or NULL)
 * all asynchronous on one stream, nothing returning to the host between frames. What the caller still fills per frame is the INPUT
 * of silk_encode_frame_FIX that is computed outside it: the samples (inputBuf after the encoder's own filters), the VAD results,
 * SNR_dB_Q7, condCoding / maxBits / useCBR, the packet's range coder, and the configuration fields (constant per encoder setting).
 * The field list is pinned on the reference by tests/test_silk_stream_cpu.py; tests/test_silk_stream_gpu.py runs streams of consecutive
 * frames with captures at t = 0 only for the state and compares every frame's payload. A fresh stream: all zero except
 * first_frame_after_reset = 1, prevLag = 100, LastGainIndex = 10 (silk_init_encoder / silk_control_encoder defaults). */
typedef struct opusgpu_silk_stream {
    int16_t x_buf[OPUSGPU_SILK_MAX_LTP_MEM + OPUSGPU_SILK_MAX_LA_SHAPE];   /* the ltp_mem_length + la_shape samples the next frame starts with */
    int16_t prev_NLSFq_Q15[OPUSGPU_SILK_MAX_ORDER];
    int32_t prevLag, prevSignalType, first_frame_after_reset, LTPCorr_Q15;
    int32_t sum_log_gain_Q7, LastGainIndex;
    int32_t HarmBoost_smth_Q16, HarmShapeGain_smth_Q16, Tilt_smth_Q16;
    int32_t ec_prevSignalType, ec_prevLagIndex, frameCounter;
    int32_t reserved[4];
} opusgpu_silk_stream;

int opusgpu_silk_stream_carry_in(const opusgpu_silk_stream *d_streams, const int16_t *d_input, const opusgpu_silk_chain_bufs *bufs,
                                 int fs_kHz, int nb_subfr, int del_dec, int n, void *hip_stream);
int opusgpu_silk_stream_carry_out(opusgpu_silk_stream *d_streams, const opusgpu_silk_chain_bufs *bufs, const opusgpu_silk_rate_ctl *d_ctl,
                                  int fs_kHz, int nb_subfr, int n, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif

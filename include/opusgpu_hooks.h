/* opusgpu_hooks.h -- the reference's per-kernel operator API (its RTCD / OVERRIDE_* hooks), per call, with the
 * reference's OWN argument lists and host pointers (SURVEY.md 8b "inner boundary").
 *
 * Every entry below is what the reference's macro or OVERRIDE_ guard for that kernel would be pointed at: same arguments,
 * same meaning, same in-place behaviour, results bit-exact. One call = one launch bracketed by small copies, so these are
 * for plumbing and parity (latency-dominated); throughput comes from the batch entry points of opusgpu.h /
 * opusgpu_silk.h, which run the same device code over N frames. Errors: opusgpu_get_last_error() (OPUSGPU_OK after a
 * successful call); on error the outputs are left untouched.
 *
 * (The hooks that existed before live in opusgpu.h / opusgpu_silk.h: opusgpu_clt_mdct_forward / _backward,
 * opusgpu_opus_fft, opusgpu_celt_pitch_xcorr, opusgpu_silk_burg_modified_c.)
 */
#ifndef OPUSGPU_HOOKS_H
#define OPUSGPU_HOOKS_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* opus_ifft(cfg, fin, fout) -- macro at opus-fix/celt/kiss_fft.h:135-178 -> opus_ifft_c (celt/kiss_fft.c:602-614):
 * unscaled inverse FFT (bit-reverse, conjugate, forward butterflies, conjugate), out of place. `cfg` must be one of the
 * four kiss_fft_state of the static 48 kHz mode (nfft 480 / 240 / 120 / 60); its head is validated, its tables are not
 * read on the device. fin / fout: kiss_fft_cpx = { int32 r, i }. */
void opusgpu_opus_ifft(const void *cfg, const void *fin, void *fout);

/* comb_filter_const(y, x, T, N, g10, g11, g12) -- guard OVERRIDE_COMB_FILTER_CONST, opus-fix/celt/celt.c:92-181
 * (comb_filter_const_c :156-181):  y[i] = x[i] + g10*x[i-T] + g11*(x[i-T+1]+x[i-T-1]) + g12*(x[i-T+2]+x[i-T-2]) in
 * MULT16_32_Q15. Reads x[-T-2 .. N-T+2) and x[0 .. N). y may equal x (the decoder's post-filter runs in place and is
 * then recursive through the already filtered samples, exactly as the C loop is); any other overlap of the two spans is
 * rejected with OPUSGPU_BAD_ARG. 1 <= N <= 8192, 3 <= T <= 4096. opus_val16 gains travel as int (C promotion). */
void opusgpu_comb_filter_const(int32_t *y, int32_t *x, int T, int N, int g10, int g11, int g12);

/* exp_rotation1(X, len, stride, c, s) -- guard OVERRIDE_vq_exp_rotation1, opus-fix/celt/vq.c:42-68: the forward and
 * backward sweep of Givens rotations over X[0..len), in place. 1 <= stride < len <= 4096. */
void opusgpu_exp_rotation1(int16_t *X, int len, int stride, int c, int s);

/* renormalise_vector(X, N, gain, arch) -- guard OVERRIDE_renormalise_vector, opus-fix/celt/vq.c:347-374: scales X to
 * norm `gain` (Q15) in place. 1 <= N <= 4096. */
void opusgpu_renormalise_vector(int16_t *X, int N, int gain, int arch);

/* silk_NSQ / silk_NSQ_del_dec with the reference's 15-argument list -- macros at opus-fix/silk/main.h:245-268 and
 * :271-296 (guards OVERRIDE_silk_NSQ / OVERRIDE_silk_NSQ_del_dec) -> silk_NSQ_c (silk/NSQ.c:74), silk_NSQ_del_dec_c
 * (silk/NSQ_del_dec.c:112). psEncC / NSQ / psIndices are the reference's own silk_encoder_state / silk_nsq_state /
 * SideInfoIndices (opus-fix/silk/structs.h; x86-64 layout): only the fields the two functions read are touched, at the
 * offsets below (pinned against the reference's headers by tests/test_hooks_layout.py); NSQ is read and written whole
 * (same layout as opusgpu_nsq_state); silk_NSQ_del_dec also writes psIndices->Seed (NSQ_del_dec.c:297). */
#define OPUSGPU_REF_SIZEOF_SILK_ENCODER_STATE 7224
#define OPUSGPU_REF_OFF_NB_SUBFR 4604               /* silk_encoder_state.nb_subfr                (opus_int) */
#define OPUSGPU_REF_OFF_FRAME_LENGTH 4608           /* .frame_length */
#define OPUSGPU_REF_OFF_SUBFR_LENGTH 4612           /* .subfr_length */
#define OPUSGPU_REF_OFF_LTP_MEM_LENGTH 4616         /* .ltp_mem_length */
#define OPUSGPU_REF_OFF_N_STATES_DEL_DEC 4652       /* .nStatesDelayedDecision */
#define OPUSGPU_REF_OFF_SHAPING_LPC_ORDER 4660      /* .shapingLPCOrder */
#define OPUSGPU_REF_OFF_PREDICT_LPC_ORDER 4664      /* .predictLPCOrder */
#define OPUSGPU_REF_OFF_WARPING_Q16 4704            /* .warping_Q16 */
#define OPUSGPU_REF_OFF_PREV_NLSFQ_Q15 4524          /* .prev_NLSFq_Q15[16]                        (opus_int16) */
#define OPUSGPU_REF_OFF_USE_INTERPOLATED_NLSFS 4656 /* .useInterpolatedNLSFs */
#define OPUSGPU_REF_OFF_FIRST_FRAME_AFTER_RESET 4696 /* .first_frame_after_reset */
#define OPUSGPU_REF_OFF_INDICES 4784                /* .indices (SideInfoIndices) */
#define OPUSGPU_REF_SIZEOF_SIDE_INFO_INDICES 36
#define OPUSGPU_REF_OFF_SIGNAL_TYPE 29              /* SideInfoIndices.signalType                 (opus_int8) */
#define OPUSGPU_REF_OFF_QUANT_OFFSET_TYPE 30        /* .quantOffsetType */
#define OPUSGPU_REF_OFF_NLSF_INTERP_COEF_Q2 31      /* .NLSFInterpCoef_Q2 */
#define OPUSGPU_REF_OFF_SEED 34                     /* .Seed */
#define OPUSGPU_REF_OFF_NLSF_INDICES 8              /* .NLSFIndices[17]                           (opus_int8) */
void opusgpu_silk_NSQ(const void *psEncC, void *NSQ, void *psIndices, const int32_t x_Q3[], int8_t pulses[],
                      const int16_t PredCoef_Q12[/*2 * 16*/], const int16_t LTPCoef_Q14[/*5 * 4*/], const int16_t AR2_Q13[/*4 * 16*/],
                      const int HarmShapeGain_Q14[/*4*/], const int Tilt_Q14[/*4*/], const int32_t LF_shp_Q14[/*4*/],
                      const int32_t Gains_Q16[/*4*/], const int pitchL[/*4*/], const int Lambda_Q10, const int LTP_scale_Q14);
void opusgpu_silk_NSQ_del_dec(const void *psEncC, void *NSQ, void *psIndices, const int32_t x_Q3[], int8_t pulses[],
                              const int16_t PredCoef_Q12[/*2 * 16*/], const int16_t LTPCoef_Q14[/*5 * 4*/], const int16_t AR2_Q13[/*4 * 16*/],
                              const int HarmShapeGain_Q14[/*4*/], const int Tilt_Q14[/*4*/], const int32_t LF_shp_Q14[/*4*/],
                              const int32_t Gains_Q16[/*4*/], const int pitchL[/*4*/], const int Lambda_Q10, const int LTP_scale_Q14);

/* quant_all_bands(...) -- opus-fix/celt/bands.c:1337-1502, the PVQ band quantiser with the tree's own 21-argument list. The
 * reference has no RTCD slot for it (plain extern, celt/bands.h): a build that wants the GPU version renames / --wraps the
 * symbol to this one. `ec` is the tree's ec_ctx INCLUDING its trailing EC_DIFF field (celt/entcode.h:63-94,
 * 56 bytes on x86-64): buf[0 .. storage) is copied to the device, the range coder runs there, and every field plus the
 * buffer come back. Supported: the static 48 kHz mode (m->Fs 48000, overlap 120, nbEBands 21), start 0, end 21,
 * LM 3, stereo (Y != NULL), storage <= 1275; anything else -> OPUSGPU_UNIMPLEMENTED in opusgpu_get_last_error() and nothing
 * is touched.
 *   encode == 1 (celt_encoder.c:2130): as in the reference's non-RESYNTH build the caller's X, Y, collapse_masks and *seed carry
 *     no information the encoder reads afterwards (celt_encoder.c:2130-2160); they are left as they were.
 *   encode == 0 (celt_decoder.c:977; bandE may be NULL as there): ec is the range DEcoder, its buffer is only read; X and Y
 *     receive the decoded normalised bands (bins [0, 800) of each channel, the bins of the 21 bands), collapse_masks[42] and
 *     *seed are written, every ec_ctx field comes back -- the decoder's lane kernel run for one stream. */
void opusgpu_quant_all_bands(int encode, const void *m, int start, int end, int16_t *X, int16_t *Y, unsigned char *collapse_masks,
                             const int32_t *bandE, int *pulses, int shortBlocks, int spread, int dual_stereo, int intensity,
                             int *tf_res, int32_t total_bits, int32_t balance, void *ec, int LM, int codedBands, uint32_t *seed,
                             int arch);

/* ec_enc_* / ec_dec_* -- opus-fix/celt/entenc.c:62-508, celt/entdec.c:93-317, celt/laplace.c:38-134. The reference has no hook
 * slot for its range coder (plain externs, celt/entenc.h, celt/entdec.h) and one symbol is a few dozen instructions, so the
 * boundary is a SCRIPT: the calls the caller would have made on `ec` (the tree's ec_enc / ec_dec incl. EC_DIFF, storage <= 1280),
 * listed as ops[n][4] = {opcode, a, b, c}, run back to back on the device in one call; `ec` (and, encoding, its buffer) comes back
 * as the reference's functions would have left it, `out[i]` receives the value decoder call i returns. Opcodes and argument
 * order: concentus_amd/csrc/ec_script.h (0 ec_encode(fl,fh,ft) 1 ec_encode_bin 2 ec_enc_bit_logp(val,logp) 3 ec_enc_uint(fl,ft)
 * 4 ec_enc_bits(fl,bits) 5 ec_enc_patch_initial_bits(val,nbits) 6 ec_enc_shrink(size) 7 ec_enc_done 8 ec_laplace_encode(value,fs,
 * decay) 9 ec_enc_icdf; 16 ec_decode(ft) 17 ec_decode_bin 18 ec_dec_update(fl,fh,ft) 19 ec_dec_bit_logp 20 ec_dec_uint
 * 21 ec_dec_bits 22 ec_laplace_decode(fs,decay) 23 ec_tell 24 ec_tell_frac 25 ec_dec_icdf). Returns OPUSGPU_OK, or
 * OPUSGPU_BAD_ARG if any argument is one the reference's celt_assert()s reject (nothing is run then). tests/test_ec_script_gpu.py
 * replays celt/tests/test_unit_entropy.c through these entry points beside the compiled reference. */
int opusgpu_ec_enc_script(void *ec, const int32_t *ops, int n_ops);
int opusgpu_ec_dec_script(void *ec, const int32_t *ops, int n_ops, int32_t *out);

/* silk_find_LPC_FIX(psEncC, NLSF_Q15, x, minInvGain_Q30) -- opus-fix/silk/fixed/find_LPC_FIX.c:37-151 (declared in
 * silk/fixed/main_FIX.h, called at silk/fixed/find_pred_coefs_FIX.c:136): reads psEncC->subfr_length / nb_subfr /
 * predictLPCOrder / useInterpolatedNLSFs / first_frame_after_reset / prev_NLSFq_Q15, writes NLSF_Q15[predictLPCOrder] and
 * psEncC->indices.NLSFInterpCoef_Q2. */
void opusgpu_silk_find_LPC_FIX(void *psEncC, int16_t NLSF_Q15[], const int16_t x[], const int32_t minInvGain_Q30);

/* silk_process_NLSFs(psEncC, PredCoef_Q12, pNLSF_Q15, prev_NLSFq_Q15) -- opus-fix/silk/process_NLSFs.c:35-106 (declared in
 * silk/main.h, called at silk/fixed/find_pred_coefs_FIX.c:139): reads psEncC->speech_activity_Q8 / nb_subfr / predictLPCOrder /
 * useInterpolatedNLSFs / NLSF_MSVQ_Survivors / indices.signalType / indices.NLSFInterpCoef_Q2 (the codebook follows
 * predictLPCOrder, as psEncC->psNLSF_CB does), quantises pNLSF_Q15 in place, writes PredCoef_Q12[2][16] and
 * psEncC->indices.NLSFIndices[predictLPCOrder + 1]. */
#define OPUSGPU_REF_OFF_SPEECH_ACTIVITY_Q8 4556     /* silk_encoder_state.speech_activity_Q8      (opus_int) */
#define OPUSGPU_REF_OFF_NLSF_MSVQ_SURVIVORS 4692    /* .NLSF_MSVQ_Survivors */
void opusgpu_silk_process_NLSFs(void *psEncC, int16_t PredCoef_Q12[/*2 * 16*/], int16_t pNLSF_Q15[], const int16_t prev_NLSFq_Q15[]);

/* silk_residual_energy_FIX(nrgs, nrgsQ, x, a_Q12, gains, subfr_length, nb_subfr, LPC_order, arch) --
 * opus-fix/silk/fixed/residual_energy_FIX.c:37-98 (declared in silk/fixed/main_FIX.h, called at
 * silk/fixed/find_pred_coefs_FIX.c:142). a_Q12 is the reference's opus_int16[2][MAX_LPC_ORDER]. */
void opusgpu_silk_residual_energy_FIX(int32_t nrgs[], int nrgsQ[], const int16_t x[], int16_t a_Q12[/*2 * 16*/], const int32_t gains[],
                                      const int subfr_length, const int nb_subfr, const int LPC_order, int arch);

/* silk_find_pred_coefs_FIX(psEnc, psEncCtrl, res_pitch, x, condCoding) -- opus-fix/silk/fixed/find_pred_coefs_FIX.c:35-148 (declared
 * in silk/fixed/main_FIX.h, called from silk_encode_frame_FIX, silk/fixed/encode_frame_FIX.c). psEnc / psEncCtrl are the
 * reference's silk_encoder_state_FIX / silk_encoder_control_FIX. Reads: sCmn.nb_subfr / subfr_length / predictLPCOrder /
 * ltp_mem_length / first_frame_after_reset / useInterpolatedNLSFs / speech_activity_Q8 / NLSF_MSVQ_Survivors / mu_LTP_Q9 /
 * LTPQuantLowComplexity / sum_log_gain_Q7 / PacketLoss_perc / nFramesPerPacket / prev_NLSFq_Q15 / indices.signalType;
 * psEncCtrl->Gains_Q16 / pitchL / coding_quality_Q14; res_pitch[0 .. ltp_mem_length + frame_length); x[-ltp_mem_length ..
 * frame_length). Writes: psEncCtrl->PredCoef_Q12 / LTPCoef_Q14 / LTPredCodGain_Q7 / ResNrg / ResNrgQ (and LTP_scale_Q14 for a
 * voiced frame); sCmn.sum_log_gain_Q7 / prev_NLSFq_Q15 / indices.NLSFIndices / NLSFInterpCoef_Q2 (and LTPIndex, PERIndex,
 * LTP_scaleIndex for a voiced frame). */
#define OPUSGPU_REF_OFF_FIX_SCMN 0                  /* silk_encoder_state_FIX.sCmn */
#define OPUSGPU_REF_OFF_MU_LTP_Q9 4684              /* silk_encoder_state.mu_LTP_Q9 */
#define OPUSGPU_REF_OFF_LTP_QUANT_LOW_COMPLEXITY 4680 /* .LTPQuantLowComplexity */
#define OPUSGPU_REF_OFF_SUM_LOG_GAIN_Q7 4688        /* .sum_log_gain_Q7 */
#define OPUSGPU_REF_OFF_PACKET_LOSS_PERC 4640       /* .PacketLoss_perc */
#define OPUSGPU_REF_OFF_N_FRAMES_PER_PACKET 5792    /* .nFramesPerPacket */
#define OPUSGPU_REF_OFF_LTP_INDEX 4                 /* SideInfoIndices.LTPIndex[4]                (opus_int8) */
#define OPUSGPU_REF_OFF_PER_INDEX 32                /* .PERIndex */
#define OPUSGPU_REF_OFF_LTP_SCALE_INDEX 33          /* .LTP_scaleIndex */
#define OPUSGPU_REF_SIZEOF_SILK_ENCODER_CONTROL_FIX 552
#define OPUSGPU_REF_OFF_CTRL_GAINS_Q16 0            /* silk_encoder_control_FIX.Gains_Q16[4]      (opus_int32) */
#define OPUSGPU_REF_OFF_CTRL_PRED_COEF_Q12 16       /* .PredCoef_Q12[2][16]                       (opus_int16) */
#define OPUSGPU_REF_OFF_CTRL_LTP_COEF_Q14 80        /* .LTPCoef_Q14[20]                           (opus_int16) */
#define OPUSGPU_REF_OFF_CTRL_LTP_SCALE_Q14 120      /* .LTP_scale_Q14                             (opus_int) */
#define OPUSGPU_REF_OFF_CTRL_PITCHL 124             /* .pitchL[4] */
#define OPUSGPU_REF_OFF_CTRL_CODING_QUALITY_Q14 484 /* .coding_quality_Q14 */
#define OPUSGPU_REF_OFF_CTRL_LTP_RED_COD_GAIN_Q7 496 /* .LTPredCodGain_Q7 */
#define OPUSGPU_REF_OFF_CTRL_RES_NRG 500            /* .ResNrg[4]                                 (opus_int32) */
#define OPUSGPU_REF_OFF_CTRL_RES_NRG_Q 516          /* .ResNrgQ[4]                                (opus_int) */
void opusgpu_silk_find_pred_coefs_FIX(void *psEnc, void *psEncCtrl, const int16_t res_pitch[], const int16_t x[], int condCoding);

/* The other four analysis calls of silk_encode_frame_FIX (opus-fix/silk/fixed/encode_frame_FIX.c:176-243) with the reference's own
 * argument lists; psEnc / psEncCtrl are the reference's silk_encoder_state_FIX / silk_encoder_control_FIX. Each reads and writes
 * exactly the fields the reference function does (listed with the batched records in include/opusgpu_silk.h), at these offsets:
 *   silk_find_pitch_lags_FIX(psEnc, psEncCtrl, res, x, arch)            silk/fixed/find_pitch_lags_FIX.c:37   (8 / 16 kHz)
 *   silk_noise_shape_analysis_FIX(psEnc, psEncCtrl, pitch_res, x, arch) silk/fixed/noise_shape_analysis_FIX.c:146
 *   silk_process_gains_FIX(psEnc, psEncCtrl, condCoding)                silk/fixed/process_gains_FIX.c:37
 *   silk_prefilter_FIX(psEnc, psEncCtrl, xw_Q3, x)                      silk/fixed/prefilter_FIX.c:102 */
#define OPUSGPU_REF_OFF_FS_KHZ 4600                                  /* silk_encoder_state.fs_kHz */
#define OPUSGPU_REF_OFF_LA_PITCH 4620                                /* silk_encoder_state.la_pitch */
#define OPUSGPU_REF_OFF_LA_SHAPE 4624                                /* silk_encoder_state.la_shape */
#define OPUSGPU_REF_OFF_SHAPE_WIN_LENGTH 4628                        /* silk_encoder_state.shapeWinLength */
#define OPUSGPU_REF_OFF_PITCH_LPC_WIN_LENGTH 4572                    /* silk_encoder_state.pitch_LPC_win_length */
#define OPUSGPU_REF_OFF_PITCH_EST_LPC_ORDER 4672                     /* silk_encoder_state.pitchEstimationLPCOrder */
#define OPUSGPU_REF_OFF_PITCH_EST_COMPLEXITY 4668                    /* silk_encoder_state.pitchEstimationComplexity */
#define OPUSGPU_REF_OFF_PITCH_EST_THRESHOLD_Q16 4676                 /* silk_encoder_state.pitchEstimationThreshold_Q16 */
#define OPUSGPU_REF_OFF_SNR_DB_Q7 4764                               /* silk_encoder_state.SNR_dB_Q7 */
#define OPUSGPU_REF_OFF_USE_CBR 4708                                 /* silk_encoder_state.useCBR */
#define OPUSGPU_REF_OFF_INPUT_QUALITY_BANDS_Q15 4744                 /* silk_encoder_state.input_quality_bands_Q15 */
#define OPUSGPU_REF_OFF_INPUT_TILT_Q15 4760                          /* silk_encoder_state.input_tilt_Q15 */
#define OPUSGPU_REF_OFF_PREV_SIGNAL_TYPE 4565                        /* silk_encoder_state.prevSignalType */
#define OPUSGPU_REF_OFF_PREV_LAG 4568                                /* silk_encoder_state.prevLag */
#define OPUSGPU_REF_OFF_GAINS_INDICES 0                              /* SideInfoIndices.GainsIndices */
#define OPUSGPU_REF_OFF_LAG_INDEX 26                                 /* SideInfoIndices.lagIndex */
#define OPUSGPU_REF_OFF_CONTOUR_INDEX 28                             /* SideInfoIndices.contourIndex */
#define OPUSGPU_REF_OFF_FIX_SSHAPE 7224                              /* silk_encoder_state_FIX.sShape */
#define OPUSGPU_REF_OFF_FIX_SPREFILT 7240                            /* silk_encoder_state_FIX.sPrefilt */
#define OPUSGPU_REF_OFF_FIX_LTPCORR_Q15 9796                         /* silk_encoder_state_FIX.LTPCorr_Q15 */
#define OPUSGPU_REF_OFF_SHAPE_LAST_GAIN_INDEX 0                      /* silk_shape_state_FIX.LastGainIndex */
#define OPUSGPU_REF_OFF_SHAPE_HARM_BOOST_SMTH_Q16 4                  /* silk_shape_state_FIX.HarmBoost_smth_Q16 */
#define OPUSGPU_REF_OFF_SHAPE_HARM_SHAPE_GAIN_SMTH_Q16 8             /* silk_shape_state_FIX.HarmShapeGain_smth_Q16 */
#define OPUSGPU_REF_OFF_SHAPE_TILT_SMTH_Q16 12                       /* silk_shape_state_FIX.Tilt_smth_Q16 */
#define OPUSGPU_REF_OFF_CTRL_AR1_Q13 140                             /* silk_encoder_control_FIX.AR1_Q13 */
#define OPUSGPU_REF_OFF_CTRL_AR2_Q13 268                             /* silk_encoder_control_FIX.AR2_Q13 */
#define OPUSGPU_REF_OFF_CTRL_LF_SHP_Q14 396                          /* silk_encoder_control_FIX.LF_shp_Q14 */
#define OPUSGPU_REF_OFF_CTRL_GAINS_PRE_Q14 412                       /* silk_encoder_control_FIX.GainsPre_Q14 */
#define OPUSGPU_REF_OFF_CTRL_HARM_BOOST_Q14 428                      /* silk_encoder_control_FIX.HarmBoost_Q14 */
#define OPUSGPU_REF_OFF_CTRL_TILT_Q14 444                            /* silk_encoder_control_FIX.Tilt_Q14 */
#define OPUSGPU_REF_OFF_CTRL_HARM_SHAPE_GAIN_Q14 460                 /* silk_encoder_control_FIX.HarmShapeGain_Q14 */
#define OPUSGPU_REF_OFF_CTRL_LAMBDA_Q10 476                          /* silk_encoder_control_FIX.Lambda_Q10 */
#define OPUSGPU_REF_OFF_CTRL_INPUT_QUALITY_Q14 480                   /* silk_encoder_control_FIX.input_quality_Q14 */
#define OPUSGPU_REF_OFF_CTRL_SPARSENESS_Q8 488                       /* silk_encoder_control_FIX.sparseness_Q8 */
#define OPUSGPU_REF_OFF_CTRL_PRED_GAIN_Q16 492                       /* silk_encoder_control_FIX.predGain_Q16 */
#define OPUSGPU_REF_OFF_CTRL_GAINS_UNQ_Q16 532                       /* silk_encoder_control_FIX.GainsUnq_Q16 */
#define OPUSGPU_REF_OFF_CTRL_LAST_GAIN_INDEX_PREV 548                /* silk_encoder_control_FIX.lastGainIndexPrev */
#define OPUSGPU_REF_SIZEOF_SILK_PREFILTER_STATE_FIX 1116
#define OPUSGPU_REF_SIZEOF_SILK_ENCODER_STATE_FIX 9800
#define OPUSGPU_REF_OFF_SVAD 32                          /* silk_encoder_state.sVAD (silk_VAD_state, 112 bytes) */
#define OPUSGPU_REF_SIZEOF_SILK_VAD_STATE 112
/* silk_VAD_GetSA_Q8_c(psEncC, pIn) -- opus-fix/silk/VAD.c:82-312 (macro silk_VAD_GetSA_Q8, silk/main.h:305-312, guard
 * OVERRIDE_silk_VAD_GetSA_Q8; called at silk/fixed/encode_frame_FIX.c:58): reads psEncC->frame_length / fs_kHz / sVAD, writes sVAD,
 * speech_activity_Q8, input_tilt_Q15, input_quality_bands_Q15[4]; returns 0. */
int opusgpu_silk_VAD_GetSA_Q8_c(void *psEncC, const int16_t pIn[]);
void opusgpu_silk_find_pitch_lags_FIX(void *psEnc, void *psEncCtrl, int16_t res[], const int16_t x[], int arch);
void opusgpu_silk_noise_shape_analysis_FIX(void *psEnc, void *psEncCtrl, const int16_t *pitch_res, const int16_t *x, int arch);
void opusgpu_silk_process_gains_FIX(void *psEnc, void *psEncCtrl, int condCoding);
void opusgpu_silk_prefilter_FIX(void *psEnc, const void *psEncCtrl, int32_t xw_Q3[], const int16_t x[]);

/* ---- silk_encode_frame_FIX as a whole, and the two entropy-coding calls inside it ----------------------------------------
 *   silk_encode_indices(psEncC, psRangeEnc, FrameIndex, encode_LBRR, condCoding)      silk/encode_indices.c:36   (encode_LBRR == 0)
 *   silk_encode_pulses(psRangeEnc, signalType, quantOffsetType, pulses, frame_length) silk/encode_pulses.c:64
 *   silk_encode_frame_FIX(psEnc, pnBytesOut, psRangeEnc, condCoding, maxBits, useCBR) silk/fixed/encode_frame_FIX.c:88
 * psRangeEnc is the tree's ec_enc (celt/entcode.h:63-94 incl. its trailing EC_DIFF), storage <= 1280. The frame hook runs the
 * reference function's sequence with every computing call replaced by the hook of the same name above and the bitrate loop's
 * decisions by opusgpu_silk_rate_control_batch (include/opusgpu_silk.h); it returns 0, or -1 with opusgpu_get_last_error() set --
 * OPUSGPU_UNIMPLEMENTED for a frame inside a bandwidth transition (sLP.mode != 0, silk_LP_variable_cutoff), with in-band LBRR, or
 * at 12 kHz (the pitch estimator's 2/3 resampler is not provided). UNIMPLEMENTED is decided before the first write to *psEnc: the
 * encoder state is untouched and a wrap shim hands the frame to __real_silk_encode_frame_FIX (INTEGRATION.md,
 * oracle/ref_gpuframe_wrap.c; the reference's caller only silk_assert()s the return value, silk/enc_API.c:499).
 * Link-level drop-in: -Wl,--wrap=silk_encode_frame_FIX; tests/test_hooks_gpu.py runs the unmodified reference
 * encoder with exactly that redirection and compares its packets with the plain reference's. */
#define OPUSGPU_REF_OFF_INPUT_BUF 5144                 /* silk_encoder_state.inputBuf */
#define OPUSGPU_REF_OFF_FRAME_COUNTER 4644             /* silk_encoder_state.frameCounter */
#define OPUSGPU_REF_OFF_PREFILL_FLAG 4712              /* silk_encoder_state.prefillFlag */
#define OPUSGPU_REF_OFF_SLP 16                         /* silk_encoder_state.sLP */
#define OPUSGPU_REF_OFF_LP_MODE 12                     /* silk_LP_state.mode */
#define OPUSGPU_REF_OFF_LBRR_ENABLED 6144              /* silk_encoder_state.LBRR_enabled */
#define OPUSGPU_REF_OFF_PULSES 4820                    /* silk_encoder_state.pulses */
#define OPUSGPU_REF_OFF_SNSQ 144                       /* silk_encoder_state.sNSQ */
#define OPUSGPU_REF_OFF_N_FRAMES_ENCODED 5796          /* silk_encoder_state.nFramesEncoded */
#define OPUSGPU_REF_OFF_EC_PREV_LAG_INDEX 5820         /* silk_encoder_state.ec_prevLagIndex (opus_int16) */
#define OPUSGPU_REF_OFF_EC_PREV_SIGNAL_TYPE 5816       /* silk_encoder_state.ec_prevSignalType */
#define OPUSGPU_REF_OFF_FIX_X_BUF 8356                 /* silk_encoder_state_FIX.x_buf */
void opusgpu_silk_encode_indices(void *psEncC, void *psRangeEnc, int FrameIndex, int encode_LBRR, int condCoding);
void opusgpu_silk_encode_pulses(void *psRangeEnc, int signalType, int quantOffsetType, int8_t pulses[], int frame_length);
int opusgpu_silk_encode_frame_FIX(void *psEnc, int32_t *pnBytesOut, void *psRangeEnc, int condCoding, int maxBits, int useCBR);

#ifdef __cplusplus
}
#endif
#endif /* OPUSGPU_HOOKS_H */

// silk_encode_frame_hook.hip -- per-call hooks, with the reference's own argument lists, for silk_encode_indices,
// silk_encode_pulses and silk_encode_frame_FIX as a whole (include/opusgpu_hooks.h): the last one is the reference function's
// sequence (opus-fix/silk/fixed/encode_frame_FIX.c:88-466) with every computing call replaced by the hook of the same name --
// pitch analysis, noise shaping analysis, prediction coefficients, gains, prefilter, the quantiser, the entropy coder -- and the
// bitrate loop's decisions taken by opusgpu_silk_rate_control_batch. What stays on the host is what the reference function does
// with memcpy / assignments: buffer shifts, the state copies of the loop, the fields carried to the next frame.
// One record per launch: plumbing / parity (a reference encoder linked against these emits the reference's packets,
// tests/test_hooks_gpu.py); throughput is the business of the batched entry points.
#include <stdlib.h>
#include <string.h>
#include <hip/hip_runtime.h>
#include "opusgpu_internal.h"
#include "../../include/opusgpu_silk.h"
#include "../../include/opusgpu_hooks.h"

namespace {

int rd_int(const void *base, int off) { int v; memcpy(&v, (const char *)base + off, sizeof(v)); return v; }
void wr_int(void *base, int off, int v) { memcpy((char *)base + off, &v, sizeof(v)); }
int16_t rd_i16(const void *base, int off) { int16_t v; memcpy(&v, (const char *)base + off, sizeof(v)); return v; }
void wr_i16(void *base, int off, int16_t v) { memcpy((char *)base + off, &v, sizeof(v)); }

// the tree's ec_ctx, x86-64 (celt/entcode.h:63-94, with the trailing EC_DIFF of this tree)
struct ref_ec_ctx {
    unsigned char *buf;
    uint32_t storage, end_offs, end_window;
    int nend_bits, nbits_total;
    uint32_t offs, rng, val, ext;
    int rem, error, EC_DIFF;
};

struct DevMem {
    void *p = nullptr;
    explicit DevMem(size_t bytes) { if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) p = nullptr; }
    ~DevMem() { if (p) (void)hipFree(p); }
    DevMem(const DevMem &) = delete;
    DevMem &operator=(const DevMem &) = delete;
};

bool ec_to_record(opusgpu_ec_state *d, const ref_ec_ctx *e)
{
    if (!e->buf || e->storage > OPUSGPU_EC_BUF || e->offs > e->storage || e->end_offs > e->storage) return false;
    memset(d, 0, sizeof(*d));
    d->storage = e->storage; d->end_offs = e->end_offs; d->end_window = e->end_window; d->nend_bits = e->nend_bits; d->nbits_total = e->nbits_total;
    d->offs = e->offs; d->rng = e->rng; d->val = e->val; d->ext = e->ext; d->rem = e->rem; d->error = e->error;
    memcpy(d->buf, e->buf, e->offs);
    memcpy(d->buf + e->storage - e->end_offs, e->buf + e->storage - e->end_offs, e->end_offs);
    return true;
}

void ec_from_record(ref_ec_ctx *e, const opusgpu_ec_state *d)
{
    e->end_offs = d->end_offs; e->end_window = d->end_window; e->nend_bits = d->nend_bits; e->nbits_total = d->nbits_total;
    e->offs = d->offs; e->rng = d->rng; e->val = d->val; e->ext = d->ext; e->rem = d->rem; e->error = d->error;
    memcpy(e->buf, d->buf, d->offs);
    memcpy(e->buf + e->storage - d->end_offs, d->buf + d->storage - d->end_offs, d->end_offs);
}

// one opusgpu_silk_bits_in record on the coder `e`
int run_bits(const opusgpu_silk_bits_in &in, ref_ec_ctx *e, opusgpu_silk_bits_out *out)
{
    opusgpu_ec_state *h_ec = (opusgpu_ec_state *)malloc(sizeof(*h_ec));
    if (!h_ec) return OPUSGPU_ALLOC_FAIL;
    OpusgpuHookBadScope bad;                 // rejected records count into this thread's counter, not the device's shared one
    int rc = ec_to_record(h_ec, e) ? bad.rc : OPUSGPU_BAD_ARG;
    DevMem din(sizeof(in)), dec(sizeof(*h_ec)), dout(sizeof(*out));
    if (rc == OPUSGPU_OK && (!din.p || !dec.p || !dout.p)) rc = OPUSGPU_ALLOC_FAIL;
    if (rc == OPUSGPU_OK && (hipMemcpy(din.p, &in, sizeof(in), hipMemcpyHostToDevice) != hipSuccess ||
                             hipMemcpy(dec.p, h_ec, sizeof(*h_ec), hipMemcpyHostToDevice) != hipSuccess))
        rc = OPUSGPU_INTERNAL_ERROR;
    if (rc == OPUSGPU_OK)
        rc = opusgpu_silk_encode_bits_batch((const opusgpu_silk_bits_in *)din.p, (opusgpu_ec_state *)dec.p, (opusgpu_silk_bits_out *)dout.p, 1, nullptr);
    if (rc == OPUSGPU_OK && (hipMemcpy(out, dout.p, sizeof(*out), hipMemcpyDeviceToHost) != hipSuccess ||
                             hipMemcpy(h_ec, dec.p, sizeof(*h_ec), hipMemcpyDeviceToHost) != hipSuccess))
        rc = OPUSGPU_INTERNAL_ERROR;
    if (rc == OPUSGPU_OK && out->status != OPUSGPU_OK) rc = out->status;
    if (rc == OPUSGPU_OK) ec_from_record(e, h_ec);
    free(h_ec);
    return rc;
}

// one step of the bitrate loop on the device: the record + the coder's ec_tell() words
int run_rate_step(opusgpu_silk_rate_ctl *ctl, const ref_ec_ctx *e)
{
    opusgpu_ec_state *h_ec = (opusgpu_ec_state *)calloc(1, sizeof(*h_ec));
    if (!h_ec) return OPUSGPU_ALLOC_FAIL;
    h_ec->storage = e->storage; h_ec->nbits_total = e->nbits_total; h_ec->rng = e->rng; h_ec->offs = e->offs;
    DevMem dctl(sizeof(*ctl)), dec(sizeof(*h_ec));
    OpusgpuHookBadScope bad;                 // rejected records count into this thread's counter, not the device's shared one
    int rc = (dctl.p && dec.p) ? bad.rc : OPUSGPU_ALLOC_FAIL;
    if (rc == OPUSGPU_OK && (hipMemcpy(dctl.p, ctl, sizeof(*ctl), hipMemcpyHostToDevice) != hipSuccess ||
                             hipMemcpy(dec.p, h_ec, sizeof(*h_ec), hipMemcpyHostToDevice) != hipSuccess))
        rc = OPUSGPU_INTERNAL_ERROR;
    if (rc == OPUSGPU_OK) rc = opusgpu_silk_rate_control_batch((opusgpu_silk_rate_ctl *)dctl.p, (const opusgpu_ec_state *)dec.p, 1, nullptr);
    if (rc == OPUSGPU_OK && hipMemcpy(ctl, dctl.p, sizeof(*ctl), hipMemcpyDeviceToHost) != hipSuccess) rc = OPUSGPU_INTERNAL_ERROR;
    if (rc == OPUSGPU_OK && ctl->status != OPUSGPU_OK) rc = ctl->status;
    free(h_ec);
    return rc;
}

void indices_to_record(opusgpu_silk_bits_in *r, const char *sCmn)
{
    const char *ind = sCmn + OPUSGPU_REF_OFF_INDICES;
    memcpy(r->GainsIndices, ind + OPUSGPU_REF_OFF_GAINS_INDICES, 4);
    memcpy(r->LTPIndex, ind + OPUSGPU_REF_OFF_LTP_INDEX, 4);
    memcpy(r->NLSFIndices, ind + OPUSGPU_REF_OFF_NLSF_INDICES, OPUSGPU_SILK_MAX_ORDER + 1);
    r->lagIndex = rd_i16(ind, OPUSGPU_REF_OFF_LAG_INDEX); r->contourIndex = (int8_t)ind[OPUSGPU_REF_OFF_CONTOUR_INDEX];
    r->signalType = (int8_t)ind[OPUSGPU_REF_OFF_SIGNAL_TYPE]; r->quantOffsetType = (int8_t)ind[OPUSGPU_REF_OFF_QUANT_OFFSET_TYPE];
    r->NLSFInterpCoef_Q2 = (int8_t)ind[OPUSGPU_REF_OFF_NLSF_INTERP_COEF_Q2]; r->PERIndex = (int8_t)ind[OPUSGPU_REF_OFF_PER_INDEX];
    r->LTP_scaleIndex = (int8_t)ind[OPUSGPU_REF_OFF_LTP_SCALE_INDEX]; r->Seed = (int8_t)ind[OPUSGPU_REF_OFF_SEED];
    r->nb_subfr = rd_int(sCmn, OPUSGPU_REF_OFF_NB_SUBFR); r->fs_kHz = rd_int(sCmn, OPUSGPU_REF_OFF_FS_KHZ);
    r->predictLPCOrder = rd_int(sCmn, OPUSGPU_REF_OFF_PREDICT_LPC_ORDER); r->frame_length = rd_int(sCmn, OPUSGPU_REF_OFF_FRAME_LENGTH);
    r->ec_prevSignalType = rd_int(sCmn, OPUSGPU_REF_OFF_EC_PREV_SIGNAL_TYPE); r->ec_prevLagIndex = rd_i16(sCmn, OPUSGPU_REF_OFF_EC_PREV_LAG_INDEX);
}

}  // namespace

// silk_encode_indices(psEncC, psRangeEnc, FrameIndex, encode_LBRR, condCoding) -- opus-fix/silk/encode_indices.c:36-183. The regular
// indices only (encode_LBRR == 0: psEncC->indices); reads nb_subfr / fs_kHz / predictLPCOrder / ec_prevSignalType / ec_prevLagIndex,
// writes the last two and the coder.
extern "C" void opusgpu_silk_encode_indices(void *psEncC, void *psRangeEnc, int FrameIndex, int encode_LBRR, int condCoding)
{
    (void)FrameIndex;
    if (!psEncC || !psRangeEnc) { opusgpu_set_last_error(OPUSGPU_BAD_ARG); return; }
    if (encode_LBRR) { opusgpu_set_last_error(OPUSGPU_UNIMPLEMENTED); return; }
    opusgpu_silk_bits_in in;
    opusgpu_silk_bits_out out;
    memset(&in, 0, sizeof(in));
    indices_to_record(&in, (const char *)psEncC);
    in.condCoding = condCoding; in.which = 1;
    const int rc = run_bits(in, (ref_ec_ctx *)psRangeEnc, &out);
    opusgpu_set_last_error(rc);
    if (rc != OPUSGPU_OK) return;
    wr_int(psEncC, OPUSGPU_REF_OFF_EC_PREV_SIGNAL_TYPE, out.ec_prevSignalType);
    wr_i16(psEncC, OPUSGPU_REF_OFF_EC_PREV_LAG_INDEX, (int16_t)out.ec_prevLagIndex);
}

// silk_encode_pulses(psRangeEnc, signalType, quantOffsetType, pulses, frame_length) -- opus-fix/silk/encode_pulses.c:64-205
extern "C" void opusgpu_silk_encode_pulses(void *psRangeEnc, int signalType, int quantOffsetType, int8_t pulses[], int frame_length)
{
    if (!psRangeEnc || !pulses || frame_length < 1 || frame_length > OPUSGPU_SILK_MAX_FRAME) { opusgpu_set_last_error(OPUSGPU_BAD_ARG); return; }
    opusgpu_silk_bits_in in;
    opusgpu_silk_bits_out out;
    memset(&in, 0, sizeof(in));
    memcpy(in.pulses, pulses, (size_t)frame_length);
    in.signalType = signalType; in.quantOffsetType = quantOffsetType; in.frame_length = frame_length; in.which = 2;
    in.nb_subfr = 4; in.fs_kHz = 16; in.predictLPCOrder = 16;                      // not read by this call; values the record check accepts
    opusgpu_set_last_error(run_bits(in, (ref_ec_ctx *)psRangeEnc, &out));
}

// silk_encode_frame_FIX(psEnc, pnBytesOut, psRangeEnc, condCoding, maxBits, useCBR) -- opus-fix/silk/fixed/encode_frame_FIX.c:88-466.
// Returns 0 like the reference; -1 with opusgpu_get_last_error() set where a stage failed or the frame needs something outside this
// path (a bandwidth-transition low-pass, silk_LP_variable_cutoff with sLP.mode != 0; in-band LBRR).
extern "C" int opusgpu_silk_encode_frame_FIX(void *psEnc, int32_t *pnBytesOut, void *psRangeEnc, int condCoding, int maxBits, int useCBR)
{
    if (!psEnc || !pnBytesOut || !psRangeEnc) { opusgpu_set_last_error(OPUSGPU_BAD_ARG); return -1; }
    char *sCmn = (char *)psEnc + OPUSGPU_REF_OFF_FIX_SCMN, *ind = sCmn + OPUSGPU_REF_OFF_INDICES, *shp = (char *)psEnc + OPUSGPU_REF_OFF_FIX_SSHAPE;
    ref_ec_ctx *e = (ref_ec_ctx *)psRangeEnc;
    const int fs_kHz = rd_int(sCmn, OPUSGPU_REF_OFF_FS_KHZ), nb_subfr = rd_int(sCmn, OPUSGPU_REF_OFF_NB_SUBFR);
    const int frame_length = rd_int(sCmn, OPUSGPU_REF_OFF_FRAME_LENGTH), ltp_mem_length = rd_int(sCmn, OPUSGPU_REF_OFF_LTP_MEM_LENGTH);
    const int la_pitch = rd_int(sCmn, OPUSGPU_REF_OFF_LA_PITCH), prefill = rd_int(sCmn, OPUSGPU_REF_OFF_PREFILL_FLAG);
    const int la_shape = 5 * fs_kHz;                                                // LA_SHAPE_MS * fs_kHz (silk/define.h)
    if (!((fs_kHz == 8 || fs_kHz == 12 || fs_kHz == 16) && (nb_subfr == 2 || nb_subfr == 4) && frame_length == 5 * fs_kHz * nb_subfr &&
          ltp_mem_length == 20 * fs_kHz && la_pitch >= 0 && la_pitch + frame_length + ltp_mem_length <= OPUSGPU_SILK_PITCH_BUF)) {
        opusgpu_set_last_error(OPUSGPU_BAD_ARG);
        return -1;
    }
    // Everything a later stage would refuse is refused HERE, before the first write to *psEnc, so that OPUSGPU_UNIMPLEMENTED
    // always means "state untouched -- the caller may run the reference's own function on this frame instead" (a wrap shim
    // does exactly that: oracle/ref_gpuframe_wrap.c): a bandwidth-transition low-pass (silk_LP_variable_cutoff, sLP.mode != 0),
    // in-band LBRR, 12 kHz (the pitch estimator's 2/3 resampler is not provided: find_pitch_lags_record_ok), and a frame
    // length the shell coder's 16-sample blocks do not divide (encode_pulses; 10 ms at 12 kHz is the only such case and is
    // already out with 12 kHz).
    if (rd_int(sCmn + OPUSGPU_REF_OFF_SLP, OPUSGPU_REF_OFF_LP_MODE) != 0 || (!prefill && rd_int(sCmn, OPUSGPU_REF_OFF_LBRR_ENABLED) != 0) ||
        fs_kHz == 12 || (frame_length & 15) != 0) {
        opusgpu_set_last_error(OPUSGPU_UNIMPLEMENTED);
        return -1;
    }
    const int fc = rd_int(sCmn, OPUSGPU_REF_OFF_FRAME_COUNTER);                     // :128
    ind[OPUSGPU_REF_OFF_SEED] = (char)(fc & 3);
    wr_int(sCmn, OPUSGPU_REF_OFF_FRAME_COUNTER, fc + 1);
    int16_t *x_buf = (int16_t *)((char *)psEnc + OPUSGPU_REF_OFF_FIX_X_BUF), *x_frame = x_buf + ltp_mem_length;
    memcpy(x_frame + la_shape, sCmn + OPUSGPU_REF_OFF_INPUT_BUF + sizeof(int16_t), sizeof(int16_t) * (size_t)frame_length);     // :145
    int rc = OPUSGPU_OK;
    if (!prefill) {
        void *ctrl = calloc(1, OPUSGPU_REF_SIZEOF_SILK_ENCODER_CONTROL_FIX);      // sEncCtrl
        int16_t *res_pitch = (int16_t *)calloc((size_t)(la_pitch + frame_length + ltp_mem_length), sizeof(int16_t));
        int32_t *xfw_Q3 = (int32_t *)calloc((size_t)frame_length, sizeof(int32_t));
        void *nsq_copy = malloc(sizeof(opusgpu_nsq_state)), *nsq_copy2 = malloc(sizeof(opusgpu_nsq_state));
        unsigned char *ec_buf_copy = (unsigned char *)malloc(OPUSGPU_EC_BUF);
        char *c = (char *)ctrl;
        if (!ctrl || !res_pitch || !xfw_Q3 || !nsq_copy || !nsq_copy2 || !ec_buf_copy) rc = OPUSGPU_ALLOC_FAIL;
#define STAGE(call) if (rc == OPUSGPU_OK) { call; rc = opusgpu_get_last_error(); }
        STAGE(opusgpu_silk_find_pitch_lags_FIX(psEnc, ctrl, res_pitch, x_frame, 0));                                             // :176
        STAGE(opusgpu_silk_noise_shape_analysis_FIX(psEnc, ctrl, res_pitch + ltp_mem_length, x_frame, 0));                       // :194
        STAGE(opusgpu_silk_find_pred_coefs_FIX(psEnc, ctrl, res_pitch, x_frame, condCoding));                                    // :210
        STAGE(opusgpu_silk_process_gains_FIX(psEnc, ctrl, condCoding));                                                          // :226
        STAGE(opusgpu_silk_prefilter_FIX(psEnc, ctrl, xfw_Q3, x_frame));                                                         // :243
        if (rc == OPUSGPU_OK) {
            // the bitrate loop (:263-423): arithmetic and decisions on the device, state copies here
            opusgpu_silk_rate_ctl ctl;
            memset(&ctl, 0, sizeof(ctl));
            ctl.maxBits = maxBits; ctl.useCBR = useCBR; ctl.condCoding = condCoding; ctl.nb_subfr = nb_subfr; ctl.frame_length = frame_length;
            for (int k = 0; k < nb_subfr; k++) {
                ctl.GainsUnq_Q16[k] = rd_int(c, OPUSGPU_REF_OFF_CTRL_GAINS_UNQ_Q16 + 4 * k);
                ctl.Gains_Q16[k] = rd_int(c, OPUSGPU_REF_OFF_CTRL_GAINS_Q16 + 4 * k);
                ctl.GainsIndices[k] = (int8_t)ind[OPUSGPU_REF_OFF_GAINS_INDICES + k];
            }
            ctl.lastGainIndexPrev = (int8_t)c[OPUSGPU_REF_OFF_CTRL_LAST_GAIN_INDEX_PREV];
            ctl.LastGainIndex = (int8_t)shp[OPUSGPU_REF_OFF_SHAPE_LAST_GAIN_INDEX];
            ctl.Lambda_Q10 = rd_int(c, OPUSGPU_REF_OFF_CTRL_LAMBDA_Q10);
            void *sNSQ = sCmn + OPUSGPU_REF_OFF_SNSQ;
            ref_ec_ctx ec_copy = *e, ec_copy2 = *e;                                // :272 (the struct, not the bytes)
            memcpy(nsq_copy, sNSQ, sizeof(opusgpu_nsq_state));
            const char seed_copy = ind[OPUSGPU_REF_OFF_SEED];
            const int16_t lag_copy = rd_i16(sCmn, OPUSGPU_REF_OFF_EC_PREV_LAG_INDEX);
            const int sig_copy = rd_int(sCmn, OPUSGPU_REF_OFF_EC_PREV_SIGNAL_TYPE);
            const int del_dec = rd_int(sCmn, OPUSGPU_REF_OFF_N_STATES_DEL_DEC) > 1 || rd_int(sCmn, OPUSGPU_REF_OFF_WARPING_Q16) > 0;
            int8_t *pulses = (int8_t *)(sCmn + OPUSGPU_REF_OFF_PULSES);
            for (int pass = 0; rc == OPUSGPU_OK; pass++) {
                if (pass > 7) { rc = OPUSGPU_INTERNAL_ERROR; break; }
                if (pass > 0) {                                                    // :283-289
                    *e = ec_copy;
                    memcpy(sNSQ, nsq_copy, sizeof(opusgpu_nsq_state));
                    ind[OPUSGPU_REF_OFF_SEED] = seed_copy;
                    wr_i16(sCmn, OPUSGPU_REF_OFF_EC_PREV_LAG_INDEX, lag_copy);
                    wr_int(sCmn, OPUSGPU_REF_OFF_EC_PREV_SIGNAL_TYPE, sig_copy);
                }
                const int16_t *PredCoef_Q12 = (const int16_t *)(c + OPUSGPU_REF_OFF_CTRL_PRED_COEF_Q12), *LTPCoef_Q14 = (const int16_t *)(c + OPUSGPU_REF_OFF_CTRL_LTP_COEF_Q14);
                const int16_t *AR2_Q13 = (const int16_t *)(c + OPUSGPU_REF_OFF_CTRL_AR2_Q13);
                const int *HarmShapeGain_Q14 = (const int *)(c + OPUSGPU_REF_OFF_CTRL_HARM_SHAPE_GAIN_Q14), *Tilt_Q14 = (const int *)(c + OPUSGPU_REF_OFF_CTRL_TILT_Q14);
                const int32_t *LF_shp_Q14 = (const int32_t *)(c + OPUSGPU_REF_OFF_CTRL_LF_SHP_Q14), *Gains_Q16 = (const int32_t *)(c + OPUSGPU_REF_OFF_CTRL_GAINS_Q16);
                const int *pitchL = (const int *)(c + OPUSGPU_REF_OFF_CTRL_PITCHL);
                const int Lambda_Q10 = rd_int(c, OPUSGPU_REF_OFF_CTRL_LAMBDA_Q10), LTP_scale_Q14 = rd_int(c, OPUSGPU_REF_OFF_CTRL_LTP_SCALE_Q14);
                if (del_dec) {                                                     // :295-306
                    STAGE(opusgpu_silk_NSQ_del_dec(sCmn, sNSQ, ind, xfw_Q3, pulses, PredCoef_Q12, LTPCoef_Q14, AR2_Q13, HarmShapeGain_Q14, Tilt_Q14,
                                                   LF_shp_Q14, Gains_Q16, pitchL, Lambda_Q10, LTP_scale_Q14));
                } else {
                    STAGE(opusgpu_silk_NSQ(sCmn, sNSQ, ind, xfw_Q3, pulses, PredCoef_Q12, LTPCoef_Q14, AR2_Q13, HarmShapeGain_Q14, Tilt_Q14, LF_shp_Q14,
                                           Gains_Q16, pitchL, Lambda_Q10, LTP_scale_Q14));
                }
                STAGE(opusgpu_silk_encode_indices(sCmn, e, rd_int(sCmn, OPUSGPU_REF_OFF_N_FRAMES_ENCODED), 0, condCoding));    // :313
                STAGE(opusgpu_silk_encode_pulses(e, (int8_t)ind[OPUSGPU_REF_OFF_SIGNAL_TYPE], (int8_t)ind[OPUSGPU_REF_OFF_QUANT_OFFSET_TYPE], pulses,
                                                 frame_length));                                                               // :320
                if (rc == OPUSGPU_OK) rc = run_rate_step(&ctl, e);
                if (rc != OPUSGPU_OK) break;
                if (ctl.save2) {                                                   // :389-395
                    ec_copy2 = *e;
                    memcpy(ec_buf_copy, e->buf, e->offs);
                    memcpy(nsq_copy2, sNSQ, sizeof(opusgpu_nsq_state));
                }
                if (ctl.restore2) {                                                // :361-368
                    *e = ec_copy2;
                    memcpy(e->buf, ec_buf_copy, ec_copy2.offs);
                    memcpy(sNSQ, nsq_copy2, sizeof(opusgpu_nsq_state));
                }
                for (int k = 0; k < nb_subfr; k++) {
                    wr_int(c, OPUSGPU_REF_OFF_CTRL_GAINS_Q16 + 4 * k, ctl.Gains_Q16[k]);
                    ind[OPUSGPU_REF_OFF_GAINS_INDICES + k] = (char)ctl.GainsIndices[k];
                }
                wr_int(c, OPUSGPU_REF_OFF_CTRL_LAMBDA_Q10, ctl.Lambda_Q10);
                shp[OPUSGPU_REF_OFF_SHAPE_LAST_GAIN_INDEX] = (char)ctl.LastGainIndex;
                if (ctl.done) break;
            }
        }
#undef STAGE
        if (rc == OPUSGPU_OK) {                                                    // :437-438
            wr_int(sCmn, OPUSGPU_REF_OFF_PREV_LAG, rd_int(c, OPUSGPU_REF_OFF_CTRL_PITCHL + 4 * (nb_subfr - 1)));
            sCmn[OPUSGPU_REF_OFF_PREV_SIGNAL_TYPE] = ind[OPUSGPU_REF_OFF_SIGNAL_TYPE];
        }
        free(ctrl); free(res_pitch); free(xfw_Q3); free(nsq_copy); free(nsq_copy2); free(ec_buf_copy);
    }
    opusgpu_set_last_error(rc);
    if (rc != OPUSGPU_OK) return -1;
    memmove(x_buf, x_buf + frame_length, sizeof(int16_t) * (size_t)(ltp_mem_length + la_shape));                                // :425
    if (prefill) { *pnBytesOut = 0; return 0; }
    wr_int(sCmn, OPUSGPU_REF_OFF_FIRST_FRAME_AFTER_RESET, 0);                      // :444
    int ilog = 0;
    for (uint32_t v = e->rng; v; v >>= 1) ilog++;
    *pnBytesOut = (e->nbits_total - ilog + 7) >> 3;                                // ec_tell (celt/entcode.h:111-113)
    return 0;
}

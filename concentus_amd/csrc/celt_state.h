// celt_state.h -- per-stream encoder state and per-batch parameters of the CELT frame path.
//
// CeltEncState is the pointer-free equivalent of everything the reference keeps between frames for one
// stream: the fields of `struct OpusCustomEncoder` past ENCODER_RESET_START plus its trailing arrays
// (opus-fix/celt/celt_encoder.c:82-128) and the Opus layer's dc_reject memory (src/opus_encoder.c:86,
// hp_mem). It lives in HBM (one record per stream); a frame kernel loads it, encodes one 20 ms frame
// and stores it back. A NULL state pointer means "every frame is the first frame of its own stream"
// (SURVEY.md 8d): the reset values below are then materialised in registers/LDS and nothing is stored.
#pragma once
#include <stdint.h>
#include "../../include/opusgpu.h"   /* opusgpu_celt_config */

#ifdef __cplusplus
extern "C" {
#endif

#define OPUSGPU_CELT_NBANDS 21
#define OPUSGPU_CELT_OVERLAP 120
#define OPUSGPU_CELT_FRAME 960
#define OPUSGPU_COMBFILTER_MAXPERIOD 1024
#define OPUSGPU_COMBFILTER_MINPERIOD 15

typedef struct opusgpu_celt_state {
    /* Opus layer (src/opus_encoder.c) */
    int32_t hp_mem[4];              /* dc_reject memories, 2 per channel */
    /* CELT layer, scalars (celt_encoder.c:82-120) */
    uint32_t rng;
    int32_t spread_decision;
    int32_t delayedIntra;
    int32_t tonal_average;
    int32_t lastCodedBands;
    int32_t hf_average;
    int32_t tapset_decision;
    int32_t prefilter_period;
    int32_t prefilter_gain;         /* opus_val16 */
    int32_t prefilter_tapset;
    int32_t consec_transient;
    int32_t preemph_memE[2];
    int32_t vbr_reservoir;
    int32_t vbr_drift;
    int32_t vbr_offset;
    int32_t vbr_count;
    int32_t overlap_max;
    int32_t stereo_saving;          /* opus_val16 */
    int32_t intensity;
    int32_t spec_avg;               /* opus_val16 */
    int32_t reserved[5];
    /* CELT layer, arrays (celt_encoder.c:123-127) */
    int16_t oldBandE[2 * OPUSGPU_CELT_NBANDS];
    int16_t oldLogE[2 * OPUSGPU_CELT_NBANDS];
    int16_t oldLogE2[2 * OPUSGPU_CELT_NBANDS];
    int16_t pad16[2];
    int32_t in_mem[2 * OPUSGPU_CELT_OVERLAP];
    int32_t prefilter_mem[2 * OPUSGPU_COMBFILTER_MAXPERIOD];
} opusgpu_celt_state;

/* Decoder counterpart: everything `struct OpusCustomDecoder` keeps past DECODER_RESET_START for a stereo
 * stream (opus-fix/celt/celt_decoder.c:66-97 and the trailing arrays :94-96; `lpc` is only used by the
 * packet loss concealment, which is not implemented). */
#define OPUSGPU_DECODE_BUFFER_SIZE 2048
typedef struct opusgpu_celt_dec_state {
    uint32_t rng;
    int32_t error;
    int32_t postfilter_period, postfilter_period_old;
    int32_t postfilter_gain, postfilter_gain_old;          /* opus_val16 */
    int32_t postfilter_tapset, postfilter_tapset_old;
    int32_t preemph_memD[2];
    int32_t loss_count;
    int32_t reserved[5];
    int16_t oldBandE[2 * OPUSGPU_CELT_NBANDS];
    int16_t oldLogE[2 * OPUSGPU_CELT_NBANDS];
    int16_t oldLogE2[2 * OPUSGPU_CELT_NBANDS];
    int16_t backgroundLogE[2 * OPUSGPU_CELT_NBANDS];
    int32_t decode_mem[2][OPUSGPU_DECODE_BUFFER_SIZE + OPUSGPU_CELT_OVERLAP];
    /* hand-off between the kernels of one opusgpu_decode_batch call (not stream state): the decoded normalised
     * bands, what the synthesis and post-filter kernels need to know about the frame, and the per-stream result */
    int16_t mid_X[2 * OPUSGPU_CELT_FRAME];
    int16_t mid_norm[2 * 624];                             /* the frame's folding source (bands.c norm / norm2): working storage of the lane kernel */
    int32_t mid_valid, mid_isTransient, mid_silence;
    int32_t mid_pf_period_old, mid_pf_period, mid_pf_period_new;
    int32_t mid_pf_gain_old, mid_pf_gain, mid_pf_gain_new;
    int32_t mid_pf_tapset_old, mid_pf_tapset, mid_pf_tapset_new;
} opusgpu_celt_dec_state;

#ifdef __cplusplus
}
#endif

// celt_enc.h -- one Opus CELT-only frame, PCM in -> packet out, by one wavefront.
//
// Restates the CELT-only slice of opus_encode_native (opus-fix/src/opus_encoder.c:938-1974: rate
// bookkeeping :1040-1054 / :1440-1453 / :1716-1770, dc_reject :1473, TOC :1925, final range :1927) and
// celt_encode_with_ec (opus-fix/celt/celt_encoder.c:1379-2273) for 48 kHz, 20 ms, restricted-lowdelay,
// fullband; the steps are numbered as in SURVEY.md 3.2.
#pragma once
#include "celt_enc_back.h"

namespace ca {

struct FrameResult { int bytes; u32 final_range; };

template <class L>
CA_DEV void load_state(FrameCtx &fc, L &F, const opusgpu_celt_state *st, int C)
{
    fc.C = C;
    if (st) {
        for (int k = 0; k < 4; k++) fc.hp_mem[k] = st->hp_mem[k];
        fc.rng = st->rng;
        fc.spread_decision = st->spread_decision;
        fc.delayedIntra = st->delayedIntra;
        fc.tonal_average = st->tonal_average;
        fc.lastCodedBands = st->lastCodedBands;
        fc.hf_average = st->hf_average;
        fc.tapset_decision = st->tapset_decision;
        fc.prefilter_period = st->prefilter_period;
        fc.prefilter_gain = st->prefilter_gain;
        fc.prefilter_tapset = st->prefilter_tapset;
        fc.consec_transient = st->consec_transient;
        fc.preemph_memE[0] = st->preemph_memE[0];
        fc.preemph_memE[1] = st->preemph_memE[1];
        fc.vbr_reservoir = st->vbr_reservoir;
        fc.vbr_drift = st->vbr_drift;
        fc.vbr_offset = st->vbr_offset;
        fc.vbr_count = st->vbr_count;
        fc.overlap_max = st->overlap_max;
        fc.stereo_saving = st->stereo_saving;
        fc.intensity = st->intensity;
        fc.spec_avg = st->spec_avg;
        fc.stereo_narrow = st->reserved[0];
        fc.hist = st->prefilter_mem;
        for (int k = lane(); k < 2 * NB; k += LANES) {
            F.oldBandE[k] = st->oldBandE[k];
            F.oldLogE[k] = st->oldLogE[k];
            F.oldLogE2[k] = st->oldLogE2[k];
        }
    } else {
        // OPUS_RESET_STATE (celt_encoder.c:2443-2462) on a zeroed encoder
        for (int k = 0; k < 4; k++) fc.hp_mem[k] = 0;
        fc.rng = 0;
        fc.spread_decision = SPREAD_NORMAL;
        fc.delayedIntra = 1;
        fc.tonal_average = 256;
        fc.lastCodedBands = 0;
        fc.hf_average = 0;
        fc.tapset_decision = 0;
        fc.prefilter_period = 0;
        fc.prefilter_gain = 0;
        fc.prefilter_tapset = 0;
        fc.consec_transient = 0;
        fc.preemph_memE[0] = fc.preemph_memE[1] = 0;
        fc.vbr_reservoir = fc.vbr_drift = fc.vbr_offset = fc.vbr_count = 0;
        fc.overlap_max = 0;
        fc.stereo_saving = 0;
        fc.intensity = 0;
        fc.spec_avg = 0;
        fc.stereo_narrow = 0;
        fc.hist = nullptr;
        for (int k = lane(); k < 2 * NB; k += LANES) {
            F.oldBandE[k] = 0;
            F.oldLogE[k] = -28672;
            F.oldLogE2[k] = -28672;
        }
    }
    wave_sync();
}

// Writes the stream state back (st_out may equal the input state: everything read from it -- the
// prefilter history in particular -- has been consumed by the time this runs).
CA_DEV void store_state_scalars(const FrameCtx &fc, opusgpu_celt_state *st)
{
    // prefilter_mem <- last 1024 samples of [history | unfiltered new]  (celt_encoder.c:1179-1187):
    // new[j] = j < 64 ? old[960 + j] : xf_unfiltered[j - 64]. The unfiltered samples were overwritten by the
    // MDCT output, so the kernel keeps them in HBM: see celt_encode_frame (hist_new).
    if (lane() == 0) {
        for (int k = 0; k < 4; k++) st->hp_mem[k] = fc.hp_mem[k];
        st->rng = fc.rng;
        st->spread_decision = fc.spread_decision;
        st->delayedIntra = fc.delayedIntra;
        st->tonal_average = fc.tonal_average;
        st->lastCodedBands = fc.lastCodedBands;
        st->hf_average = fc.hf_average;
        st->tapset_decision = fc.tapset_decision;
        st->prefilter_period = fc.prefilter_period;
        st->prefilter_gain = fc.prefilter_gain;
        st->prefilter_tapset = fc.prefilter_tapset;
        st->consec_transient = fc.consec_transient;
        st->preemph_memE[0] = fc.preemph_memE[0];
        st->preemph_memE[1] = fc.preemph_memE[1];
        st->vbr_reservoir = fc.vbr_reservoir;
        st->vbr_drift = fc.vbr_drift;
        st->vbr_offset = fc.vbr_offset;
        st->vbr_count = fc.vbr_count;
        st->overlap_max = fc.overlap_max;
        st->stereo_saving = fc.stereo_saving;
        st->intensity = fc.intensity;
        st->spec_avg = fc.spec_avg;
        st->reserved[0] = fc.stereo_narrow;
    }
}

#if !defined(CA_LANE_FRAME)
template <class L>
CA_DEV void store_state(const FrameCtx &fc, L &F, opusgpu_celt_state *st)
{
    const int C = fc.C;
    store_state_scalars(fc, st);
    for (int k = lane(); k < C * NB; k += LANES) {
        st->oldBandE[k] = F.oldBandE[k];
        st->oldLogE[k] = F.oldLogE[k];
        st->oldLogE2[k] = F.oldLogE2[k];
    }
}
#endif

// ---- hand-off record between the two kernels (HBM, one per frame in flight, pointer-free) ----------
struct FrameMid {
    // range coder state at the hand-off (RangeEnc without its buffer pointer) + the bytes emitted so far
    u32 ec_storage, ec_end_offs, ec_end_window, ec_offs, ec_rng, ec_val, ec_ext;
    i32 ec_nend_bits, ec_nbits_total, ec_rem, ec_error;
    // frame-level decisions and budgets
    i32 max_data_bytes, nbCompressedBytes, nbAvailableBytes, vbr_rate, effectiveBytes, equiv_rate, total_bits;
    i32 silence, pitch_index, gain1, pf_on, prefilter_tapset;
    i32 isTransient, shortBlocks, tf_chan, tf_estimate, transient_got_disabled, temporal_vbr;
    // stream state scalars as of the hand-off (FrameCtx)
    i32 hp_mem[4];
    u32 rng;
    i32 spread_decision, delayedIntra, tonal_average, lastCodedBands, hf_average, tapset_decision;
    i32 prefilter_period, prefilter_gain, prefilter_tapset_state, consec_transient;
    i32 preemph_memE[2];
    i32 vbr_reservoir, vbr_drift, vbr_offset, vbr_count, overlap_max, stereo_saving, intensity, spec_avg;
    i32 trans_unmask[2];           // per-channel masking metric (transient stage kernel; split pipeline only)
    i32 pad[4];                    // keeps X 16-byte aligned (it is moved with 128-bit accesses)
    u8 packet_head[32];
    i32 bandE[2 * NB];
    i16 bandLogE[2 * NB], bandLogE2[2 * NB], oldBandE[2 * NB], oldLogE[2 * NB], oldLogE2[2 * NB];
    i16 pad16[2];
    i16 X[2 * FRAME];
};

}  // namespace ca
#if defined(CA_LANE_FRAME)
#include "celt_enc_lane.h"
#endif
namespace ca {

// Scalars that travel between the phases next to the range-coder state and the FrameCtx.
struct MidScalars {
    int max_data_bytes, nbCompressedBytes, nbAvailableBytes, effectiveBytes;
    i32 vbr_rate, equiv_rate, total_bits;
    int silence, pitch_index, pf_on, prefilter_tapset;
    i32 gain1;
    int isTransient, shortBlocks, tf_chan, transient_got_disabled;
    i32 tf_estimate, temporal_vbr;
};

template <class L>
CA_DEV void mid_store(L &F, FrameMid *mid, const RangeEnc &enc, const FrameCtx &fc, const MidScalars &m)
{
    for (int k = lane(); k < 2 * NB; k += LANES) {
        mid->bandE[k] = F.bandE[k];
        mid->bandLogE[k] = F.bandLogE[k];
        mid->bandLogE2[k] = F.bandLogE2[k];
        mid->oldBandE[k] = F.oldBandE[k];
        mid->oldLogE[k] = F.oldLogE[k];
        mid->oldLogE2[k] = F.oldLogE2[k];
    }
    for (int k = lane(); k < 32; k += LANES) mid->packet_head[k] = F.packet[1 + k];
    if (lane() == 0) {
        mid->ec_storage = enc.storage; mid->ec_end_offs = enc.end_offs; mid->ec_end_window = enc.end_window;
        mid->ec_offs = enc.offs; mid->ec_rng = enc.rng; mid->ec_val = enc.val; mid->ec_ext = enc.ext;
        mid->ec_nend_bits = enc.nend_bits; mid->ec_nbits_total = enc.nbits_total; mid->ec_rem = enc.rem;
        mid->ec_error = (enc.offs > 32 || enc.end_offs > 0) ? -1 : enc.error;
        mid->max_data_bytes = m.max_data_bytes; mid->nbCompressedBytes = m.nbCompressedBytes;
        mid->nbAvailableBytes = m.nbAvailableBytes; mid->vbr_rate = m.vbr_rate; mid->effectiveBytes = m.effectiveBytes;
        mid->equiv_rate = m.equiv_rate; mid->total_bits = m.total_bits;
        mid->silence = m.silence; mid->pitch_index = m.pitch_index; mid->gain1 = m.gain1; mid->pf_on = m.pf_on;
        mid->prefilter_tapset = m.prefilter_tapset;
        mid->isTransient = m.isTransient; mid->shortBlocks = m.shortBlocks; mid->tf_chan = m.tf_chan;
        mid->tf_estimate = m.tf_estimate; mid->transient_got_disabled = m.transient_got_disabled;
        mid->temporal_vbr = m.temporal_vbr;
        for (int k = 0; k < 4; k++) mid->hp_mem[k] = fc.hp_mem[k];
        mid->rng = fc.rng; mid->spread_decision = fc.spread_decision; mid->delayedIntra = fc.delayedIntra;
        mid->tonal_average = fc.tonal_average; mid->lastCodedBands = fc.lastCodedBands; mid->hf_average = fc.hf_average;
        mid->tapset_decision = fc.tapset_decision; mid->prefilter_period = fc.prefilter_period;
        mid->prefilter_gain = fc.prefilter_gain; mid->prefilter_tapset_state = fc.prefilter_tapset;
        mid->consec_transient = fc.consec_transient;
        mid->preemph_memE[0] = fc.preemph_memE[0]; mid->preemph_memE[1] = fc.preemph_memE[1];
        mid->vbr_reservoir = fc.vbr_reservoir; mid->vbr_drift = fc.vbr_drift; mid->vbr_offset = fc.vbr_offset;
        mid->vbr_count = fc.vbr_count; mid->overlap_max = fc.overlap_max; mid->stereo_saving = fc.stereo_saving;
        mid->intensity = fc.intensity; mid->spec_avg = fc.spec_avg;
        mid->pad[0] = fc.stereo_narrow;
    }
}

template <class L>
CA_DEV void mid_load(L &F, const FrameMid *mid, RangeEnc &enc, FrameCtx &fc, MidScalars &m, int C)
{
    fc.C = C;
    fc.hist = nullptr;
    for (int k = 0; k < 4; k++) fc.hp_mem[k] = uni(mid->hp_mem[k]);
    fc.rng = uni(mid->rng); fc.spread_decision = uni(mid->spread_decision); fc.delayedIntra = uni(mid->delayedIntra);
    fc.tonal_average = uni(mid->tonal_average); fc.lastCodedBands = uni(mid->lastCodedBands); fc.hf_average = uni(mid->hf_average);
    fc.tapset_decision = uni(mid->tapset_decision); fc.prefilter_period = uni(mid->prefilter_period);
    fc.prefilter_gain = uni(mid->prefilter_gain); fc.prefilter_tapset = uni(mid->prefilter_tapset_state);
    fc.consec_transient = uni(mid->consec_transient);
    fc.preemph_memE[0] = uni(mid->preemph_memE[0]); fc.preemph_memE[1] = uni(mid->preemph_memE[1]);
    fc.vbr_reservoir = uni(mid->vbr_reservoir); fc.vbr_drift = uni(mid->vbr_drift); fc.vbr_offset = uni(mid->vbr_offset);
    fc.vbr_count = uni(mid->vbr_count); fc.overlap_max = uni(mid->overlap_max); fc.stereo_saving = uni(mid->stereo_saving);
    fc.intensity = uni(mid->intensity); fc.spec_avg = uni(mid->spec_avg);
    fc.stereo_narrow = uni(mid->pad[0]);
    m.max_data_bytes = uni(mid->max_data_bytes); m.nbCompressedBytes = uni(mid->nbCompressedBytes);
    m.nbAvailableBytes = uni(mid->nbAvailableBytes); m.vbr_rate = uni(mid->vbr_rate); m.effectiveBytes = uni(mid->effectiveBytes);
    m.equiv_rate = uni(mid->equiv_rate); m.total_bits = uni(mid->total_bits);
    m.silence = uni(mid->silence); m.pitch_index = uni(mid->pitch_index); m.gain1 = uni(mid->gain1); m.pf_on = uni(mid->pf_on);
    m.prefilter_tapset = uni(mid->prefilter_tapset);
    m.isTransient = uni(mid->isTransient); m.shortBlocks = uni(mid->shortBlocks); m.tf_chan = uni(mid->tf_chan);
    m.tf_estimate = uni(mid->tf_estimate); m.transient_got_disabled = uni(mid->transient_got_disabled);
    m.temporal_vbr = uni(mid->temporal_vbr);
    enc.buf = F.packet + 1;
    enc.storage = uni(mid->ec_storage); enc.end_offs = uni(mid->ec_end_offs); enc.end_window = uni(mid->ec_end_window);
    enc.offs = uni(mid->ec_offs); enc.rng = uni(mid->ec_rng); enc.val = uni(mid->ec_val); enc.ext = uni(mid->ec_ext);
    enc.nend_bits = uni(mid->ec_nend_bits); enc.nbits_total = uni(mid->ec_nbits_total); enc.rem = uni(mid->ec_rem);
    enc.error = uni(mid->ec_error);
    for (int k = lane(); k < 2 * NB; k += LANES) {
        F.bandE[k] = mid->bandE[k];
        F.bandLogE[k] = mid->bandLogE[k];
        F.bandLogE2[k] = mid->bandLogE2[k];
        F.oldBandE[k] = mid->oldBandE[k];
        F.oldLogE[k] = mid->oldLogE[k];
        F.oldLogE2[k] = mid->oldLogE2[k];
    }
    for (int k = lane(); k < 32; k += LANES) F.packet[1 + k] = mid->packet_head[k];
    wave_sync();
}

// Phase 1: PCM -> FrameMid (steps 0-10 of SURVEY 3.2). pcm: 960*C interleaved int16 (global).
// st_in == nullptr: independent first frame. st_out may be nullptr or == st_in; this phase stores the
// time-domain part of the stream state (in_mem, prefilter_mem), phase 2 stores the rest.
// PHASE 0: the whole phase in one call. The split pipeline runs it as two kernels around the serial
// stages that have their own lane-per-channel kernels (celt_stage_kernels.hip):
//   PHASE 1: dc_reject already done (planar int16 in mid->X, filter memory in mid->hp_mem); runs up to the
//            pitch pre-filter and parks the time signal in in_ws ([2][1080] int32) + the scalars in mid;
//   PHASE 2: resumes there with the transient metric of mid->trans_unmask.
template <int PHASE, class L>
CA_DEVFN void celt_encode_front_phase(L &F, const opusgpu_celt_config &cfg, const opusgpu_celt_state *st_in,
                                      opusgpu_celt_state *st_out, const i16 *pcm, FrameMid *mid, StageClock *stage_clock,
                                      i32 *in_ws)
{
    (void)stage_clock;
    if (lane() == 0) F.diag = (void *)stage_clock;
    wave_sync();
    // stereo is the only channel count the entry points admit (config_ok): a constant, so that the channel loops unroll
    const int C = 2, N = FRAME, LM = LM3, M = M8, end = NB;
    (void)cfg.channels;
    FrameCtx fc;
    RangeEnc enc;
    int max_data_bytes, nbCompressedBytes, nbAvailableBytes, effectiveBytes, silence, pitch_index, pf_on, prefilter_tapset;
    i32 vbr_rate, equiv_rate, total_bits, gain1;
  if constexpr (PHASE != 2) {
    load_state(fc, F, st_in, C);

    // ---- Opus layer: rate bookkeeping (opus_encoder.c:1040-1054, 1440-1453, 1735-1770, 1873-1876) ----
    max_data_bytes = imin(1276, cfg.max_data_bytes);
    i32 bitrate_bps = cfg.bitrate;
    if (!cfg.vbr) {
        int cbrBytes = imin((3 * bitrate_bps / 8 + 150 / 2) / 150, max_data_bytes);
        bitrate_bps = cbrBytes * 150 * 8 / 3;
        max_data_bytes = cbrBytes;
    }
    const int bytes_target = imin(max_data_bytes, bitrate_bps * N / (48000 * 8)) - 1;
    int nb_compr_bytes = cfg.vbr ? max_data_bytes - 1 : bytes_target;
    nb_compr_bytes = imin(max_data_bytes - 1, nb_compr_bytes);
    // CELT ctl: VBR on -> bitrate = bitrate_bps, else OPUS_BITRATE_MAX (-1) with vbr off (opus_encoder.c:1727-1768)
    const int celt_vbr = cfg.vbr;
    const i32 celt_bitrate = cfg.vbr ? bitrate_bps : -1;
    const int constrained_vbr = cfg.constrained_vbr;

    ec_enc_init(enc, F.packet + 1, (u32)(max_data_bytes - 1));
    ec_enc_shrink(enc, (u32)nb_compr_bytes);

    // ---- stage PCM, dc_reject (opus_encoder.c:1473) ----
    if constexpr (PHASE == 1) {
        const int4 *src = reinterpret_cast<const int4 *>(mid->X);
        int4 *dst = reinterpret_cast<int4 *>(frame_pcmf(F));
        for (int k = lane(); k < (N * C * 2) / 16; k += LANES) dst[k] = src[k];
        for (int k = 0; k < 4; k++) fc.hp_mem[k] = uni(mid->hp_mem[k]);
        wave_sync();
    } else {
        const int4 *src = reinterpret_cast<const int4 *>(pcm);
        int4 *dst = reinterpret_cast<int4 *>(F.s.raw_pcm);
        for (int k = lane(); k < (N * C * 2) / 16; k += LANES) dst[k] = src[k];
        wave_sync();
        dc_reject_wave(F, fc);
    }
    stereo_width_wave(F, fc, bitrate_bps);
    CA_TRACE("dc_reject done");

    CA_STAMP(0);
    // ---- celt_encode_with_ec ----
    i32 tell = ec_tell(enc);                                   // == 1
    const int nbFilledBytes = (tell + 4) >> 3;                 // == 0
    nbCompressedBytes = imin(nb_compr_bytes, 1275);
    nbAvailableBytes = nbCompressedBytes - nbFilledBytes;
    if (celt_vbr && celt_bitrate != -1) {
        i32 den = 48000 >> BITRES;
        vbr_rate = (celt_bitrate * N + (den >> 1)) / den;
        effectiveBytes = vbr_rate >> (3 + BITRES);
    } else {
        vbr_rate = 0;
        i32 tmp = celt_bitrate * N;
        if (tell > 1) tmp += tell;
        if (celt_bitrate != -1) nbCompressedBytes = imax(2, imin(nbCompressedBytes, (tmp + 4 * 48000) / (8 * 48000)));
        effectiveBytes = nbCompressedBytes;
    }
    equiv_rate = 510000;
    if (celt_bitrate != -1) equiv_rate = celt_bitrate - (40 * C + 20) * ((400 >> LM) - 50);
    if (vbr_rate > 0 && constrained_vbr) {
        i32 vbr_bound = vbr_rate;
        i32 max_allowed = imin(imax(tell == 1 ? 2 : 0, (vbr_rate + vbr_bound - fc.vbr_reservoir) >> (BITRES + 3)), nbAvailableBytes);
        if (max_allowed < nbAvailableBytes) {
            nbCompressedBytes = nbFilledBytes + max_allowed;
            nbAvailableBytes = max_allowed;
            ec_enc_shrink(enc, (u32)nbCompressedBytes);
        }
    }
    total_bits = nbCompressedBytes * 8;

    // 2. silence
    i32 sample_max = imax(fc.overlap_max, maxabs_pcm(F, C, 0, N - OVL));
    fc.overlap_max = maxabs_pcm(F, C, N - OVL, N);
    sample_max = imax(sample_max, fc.overlap_max);
    silence = sample_max == 0;
    if (tell == 1) ec_enc_bit_logp(enc, silence, 15);
    else silence = 0;
    if (silence) {
        if (vbr_rate > 0) {
            effectiveBytes = nbCompressedBytes = imin(nbCompressedBytes, nbFilledBytes + 2);
            total_bits = nbCompressedBytes * 8;
            nbAvailableBytes = 2;
            ec_enc_shrink(enc, (u32)nbCompressedBytes);
        }
        tell = nbCompressedBytes * 8;
        enc.nbits_total += tell - ec_tell(enc);
    }

    CA_STAMP(1);
    // 3. pre-emphasis
    preemphasis_wave(F, fc);
    if constexpr (!L::IN_IS_GLOBAL) CA_TAP("in_preemph", F.in, sizeof(F.in));
    CA_TRACE("preemph done");

    CA_STAMP(2);
    // 4. pitch pre-filter
    {
        const int enabled = (nbAvailableBytes > 12 * C) && !silence && cfg.complexity >= 5;
        prefilter_tapset = fc.tapset_decision;
        PrefilterOut po = run_prefilter_wave(F, fc, st_in ? st_in->in_mem : nullptr, prefilter_tapset, enabled,
                                             nbAvailableBytes, cfg.loss_rate);
        pitch_index = po.pitch_index;
        gain1 = po.gain1;
        pf_on = po.pf_on;
        if (pf_on == 0) {
            if (tell + 16 <= total_bits) ec_enc_bit_logp(enc, 0, 1);
        } else {
            ec_enc_bit_logp(enc, 1, 1);
            pitch_index += 1;
            int octave = ec_ilog((u32)pitch_index) - 5;
            ec_enc_uint(enc, (u32)octave, 6);
            ec_enc_bits(enc, (u32)(pitch_index - (16 << octave)), (u32)(4 + octave));
            pitch_index -= 1;
            ec_enc_bits(enc, (u32)po.qg, 3);
            ec_enc_icdf(enc, prefilter_tapset, CLT_tapset_icdf, 2);
        }
    }
    // stream state that depends on the (still live) unfiltered / filtered time signal
    if (st_out) {
        for (int c = 0; c < C; c++) {
            for (int i = lane(); i < OVL; i += LANES) st_out->in_mem[c * OVL + i] = tsig(F, c)[N + i];
            // prefilter_mem <- last 1024 of [history | new]; new[j] = j<64 ? old[960+j] : unfiltered[j-64].
            // In-place safe: element j (>= 64) no longer depends on the old array, element j < 64 reads old[960+j]
            // which only lanes handling j' = 960+j >= 64 overwrite -> read everything first.
            i32 *keep = F.s.pitch.xcorr;                     // pitch scratch is dead by now
            for (int j = lane(); j < MAXP - FRAME; j += LANES) keep[j] = fc.hist ? fc.hist[c * MAXP + FRAME + j] : 0;
            wave_sync();
            for (int j = lane(); j < MAXP; j += LANES)
                st_out->prefilter_mem[c * MAXP + j] = j < MAXP - FRAME ? keep[j] : F.xf[c][j - (MAXP - FRAME)];
            wave_sync();
        }
    }
    wave_sync();

    CA_TRACE("prefilter done pitch=%d gain=%d pf_on=%d", pitch_index, gain1, pf_on); CA_TRACE("");
    if constexpr (!L::IN_IS_GLOBAL) CA_TAP("in_filtered", F.in, sizeof(F.in));
    CA_STAMP(3);
    if constexpr (PHASE == 1) {
        if (!L::IN_IS_GLOBAL)
            for (int c = 0; c < C; c++)
                for (int i = lane(); i < N + OVL; i += LANES) in_ws[c * (N + OVL) + i] = tsig(F, c)[i];
        MidScalars m;
        m.max_data_bytes = max_data_bytes; m.nbCompressedBytes = nbCompressedBytes; m.nbAvailableBytes = nbAvailableBytes;
        m.effectiveBytes = effectiveBytes; m.vbr_rate = vbr_rate; m.equiv_rate = equiv_rate; m.total_bits = total_bits;
        m.silence = silence; m.pitch_index = pitch_index; m.pf_on = pf_on; m.prefilter_tapset = prefilter_tapset; m.gain1 = gain1;
        m.isTransient = m.shortBlocks = m.tf_chan = m.transient_got_disabled = 0;
        m.tf_estimate = m.temporal_vbr = 0;
        mid_store(F, mid, enc, fc, m);
        return;
    }
  } else {
    MidScalars m;
    mid_load(F, mid, enc, fc, m, C);
    max_data_bytes = m.max_data_bytes; nbCompressedBytes = m.nbCompressedBytes; nbAvailableBytes = m.nbAvailableBytes;
    effectiveBytes = m.effectiveBytes; vbr_rate = m.vbr_rate; equiv_rate = m.equiv_rate; total_bits = m.total_bits;
    silence = m.silence; pitch_index = m.pitch_index; pf_on = m.pf_on; prefilter_tapset = m.prefilter_tapset; gain1 = m.gain1;
    if (!L::IN_IS_GLOBAL)
        for (int c = 0; c < C; c++)
            for (int i = lane(); i < N + OVL; i += LANES) tsig(F, c)[i] = in_ws[c * (N + OVL) + i];
    wave_sync();
  }
  if constexpr (PHASE != 1) {
    // 5. transient analysis
    int isTransient = 0, shortBlocks = 0, tf_chan = 0, transient_got_disabled = 0;
    i32 tf_estimate = 0;
    if (cfg.complexity >= 1) {
        TransientOut to;
        if constexpr (PHASE == 2) {
            i32 um[2] = {uni(mid->trans_unmask[0]), uni(mid->trans_unmask[1])};
            to = transient_combine(um, C);
        } else {
            to = transient_analysis_wave(F, fc);
        }
        isTransient = to.is_transient;
        tf_estimate = to.tf_estimate;
        tf_chan = to.tf_chan;
    }
    if (ec_tell(enc) + 3 <= total_bits) {
        if (isTransient) shortBlocks = M;
    } else {
        isTransient = 0;
        transient_got_disabled = 1;
    }

    CA_TRACE("transient done isT=%d tf_est=%d", isTransient, tf_estimate); CA_TRACE("");
    CA_STAMP(4);
    // 6./7./8./9. MDCT + band energies, temporal VBR, transient patch.
    // The reference runs compute_mdcts up to three times (extra long-block pass for bandLogE2 when
    // complexity >= 8, the main pass, and a short-block redo when patch_transient_decision fires,
    // celt_encoder.c:1698-1711, :1825-1845); here they are three trips through ONE inlined copy.
    const int secondMdct = shortBlocks && cfg.complexity >= 8;
    i32 temporal_vbr = 0;
    for (int mdct_pass = secondMdct ? 0 : 1;;) {
        CA_STAMP(5);
        if constexpr (L::XF_CHANNELS == 1) {
            // one coefficient buffer: a channel is transformed, measured and (except in the long-block pre-pass, whose only
            // product is bandLogE2) normalised before the next one takes the buffer. Normalising here is speculative -- the
            // transient patch below may order the frame transformed again, which simply repeats all of this with short blocks --
            // and exact: a band's gain depends on that channel's own band energy only (bands.c:146-168).
            for (int c = 0; c < C; c++) {
                compute_mdct_channel(F, c, mdct_pass == 0 ? 0 : shortBlocks);
                band_energies_channel(F, c, mdct_pass == 0 ? F.bandLogE2 : F.bandLogE);
                if (mdct_pass != 0) normalise_bands_channel(F, c);
            }
        } else {
            compute_mdcts_wave(F, fc, mdct_pass == 0 ? 0 : shortBlocks);
            CA_STAMP(30);
            band_energies_wave(F, fc, mdct_pass == 0 ? F.bandLogE2 : F.bandLogE);
        }
        CA_STAMP(31);
        if (mdct_pass == 0) {
            for (int k = lane(); k < C * NB; k += LANES) F.bandLogE2[k] = (i16)(F.bandLogE2[k] + (shl16(LM, 10) >> 1));
            wave_sync();
            mdct_pass = 1;
            continue;
        }
        if (mdct_pass == 2) {
            for (int k = lane(); k < C * NB; k += LANES) F.bandLogE2[k] = (i16)(F.bandLogE2[k] + (shl16(LM, 10) >> 1));
            wave_sync();
            tf_estimate = 3277;                                  // QCONST16(.2f,14)
            break;
        }
        if constexpr (L::XF_CHANNELS != 1) CA_TAP("freq", F.xf, sizeof(F.xf));
        CA_TAP("bandE", F.bandE, sizeof(F.bandE)); CA_TAP("bandLogE", F.bandLogE, sizeof(F.bandLogE));
        {   // temporal VBR (celt_encoder.c:1803-1819)
            i32 follow = -10240;
            i32 frame_avg = 0;
            i32 offset = shortBlocks ? (shl16(LM, 10) >> 1) : 0;
            for (int i = 0; i < end; i++) {
                follow = (i16)imax(follow - 1024, F.bandLogE[i] - offset);
                if (C == 2) follow = (i16)imax(follow, F.bandLogE[i + NB] - offset);
                frame_avg += follow;
            }
            frame_avg /= end;
            temporal_vbr = (i16)sub16(frame_avg, fc.spec_avg);
            temporal_vbr = imin(3072, imax(-1536, temporal_vbr));
            fc.spec_avg = (i16)(fc.spec_avg + mul16_16_q15(655, temporal_vbr));
        }
        if (!secondMdct) {
            for (int k = lane(); k < C * NB; k += LANES) F.bandLogE2[k] = F.bandLogE[k];
            wave_sync();
        }
        if (!(ec_tell(enc) + 3 <= total_bits && !isTransient && cfg.complexity >= 5)) break;
        // patch_transient_decision (celt_encoder.c:380-416); spread_old lives in LDS (F.follower is free here)
        i16 *spread_old = F.follower;
        if (lane() == 0) {
            if (C == 1) {
                spread_old[0] = F.oldBandE[0];
                for (int i = 1; i < end; i++) spread_old[i] = (i16)imax(spread_old[i - 1] - 1024, F.oldBandE[i]);
            } else {
                spread_old[0] = (i16)imax(F.oldBandE[0], F.oldBandE[NB]);
                for (int i = 1; i < end; i++)
                    spread_old[i] = (i16)imax(spread_old[i - 1] - 1024, imax(F.oldBandE[i], F.oldBandE[i + NB]));
            }
            for (int i = end - 2; i >= 0; i--) spread_old[i] = (i16)imax(spread_old[i], spread_old[i + 1] - 1024);
        }
        wave_sync();
        i32 mean_diff = 0;
        for (int c = 0; c < C; c++)
            for (int i = 2; i < end - 1; i++) {
                i32 x1 = imax(0, F.bandLogE[i + c * NB]), x2 = imax(0, spread_old[i]);
                mean_diff = add32(mean_diff, imax(0, sub16(x1, x2)));
            }
        mean_diff = mean_diff / (C * (end - 1 - 2));
        wave_sync();
        if (!(mean_diff > 1024)) break;
        isTransient = 1;
        shortBlocks = M;
        mdct_pass = 2;
    }
    if (ec_tell(enc) + 3 <= total_bits) ec_enc_bit_logp(enc, isTransient, 3);

    CA_STAMP(5);
    // 10. normalise
    if constexpr (L::XF_CHANNELS != 1) normalise_bands_wave(F, fc);

    if constexpr (!L::IN_IS_GLOBAL) CA_TAP("X", F.in, 2 * FRAME * 2);
    CA_STAMP(6);
    // ---- hand-off ----
    {
        const i16 *X = frame_X(F);
        if (X != mid->X)
            for (int k = lane(); k < 2 * FRAME; k += LANES) mid->X[k] = X[k];
        MidScalars m;
        m.max_data_bytes = max_data_bytes; m.nbCompressedBytes = nbCompressedBytes; m.nbAvailableBytes = nbAvailableBytes;
        m.effectiveBytes = effectiveBytes; m.vbr_rate = vbr_rate; m.equiv_rate = equiv_rate; m.total_bits = total_bits;
        m.silence = silence; m.pitch_index = pitch_index; m.pf_on = pf_on; m.prefilter_tapset = prefilter_tapset; m.gain1 = gain1;
        m.isTransient = isTransient; m.shortBlocks = shortBlocks; m.tf_chan = tf_chan;
        m.transient_got_disabled = transient_got_disabled; m.tf_estimate = tf_estimate; m.temporal_vbr = temporal_vbr;
        mid_store(F, mid, enc, fc, m);
    }
  }
}

template <class L>
CA_DEVFN void celt_encode_front(L &F, const opusgpu_celt_config &cfg, const opusgpu_celt_state *st_in,
                                opusgpu_celt_state *st_out, const i16 *pcm, FrameMid *mid, StageClock *stage_clock = nullptr)
{
    celt_encode_front_phase<0>(F, cfg, st_in, st_out, pcm, mid, stage_clock, nullptr);
}

// Phase 2: FrameMid -> packet (steps 11-19). out: packet bytes (global, >= max packet size).
template <class L>
CA_DEVFN FrameResult celt_encode_back(L &F, const opusgpu_celt_config &cfg, const FrameMid *mid, opusgpu_celt_state *st_out,
                                      u8 *out, StageClock *stage_clock = nullptr)
{
    (void)stage_clock;
    if (lane() == 0) F.diag = (void *)stage_clock;
    const int C = 2, LM = LM3, end = NB;                                           // stereo only (config_ok): a constant
    const int celt_vbr = cfg.vbr, constrained_vbr = cfg.constrained_vbr;
    const int nbFilledBytes = 0;
    FrameCtx fc;
    fc.C = C;
    fc.hist = nullptr;
    for (int k = 0; k < 4; k++) fc.hp_mem[k] = uni(mid->hp_mem[k]);
    fc.rng = uni(mid->rng); fc.spread_decision = uni(mid->spread_decision); fc.delayedIntra = uni(mid->delayedIntra);
    fc.tonal_average = uni(mid->tonal_average); fc.lastCodedBands = uni(mid->lastCodedBands); fc.hf_average = uni(mid->hf_average);
    fc.tapset_decision = uni(mid->tapset_decision); fc.prefilter_period = uni(mid->prefilter_period);
    fc.prefilter_gain = uni(mid->prefilter_gain); fc.prefilter_tapset = uni(mid->prefilter_tapset_state);
    fc.consec_transient = uni(mid->consec_transient);
    fc.preemph_memE[0] = uni(mid->preemph_memE[0]); fc.preemph_memE[1] = uni(mid->preemph_memE[1]);
    fc.vbr_reservoir = uni(mid->vbr_reservoir); fc.vbr_drift = uni(mid->vbr_drift); fc.vbr_offset = uni(mid->vbr_offset);
    fc.vbr_count = uni(mid->vbr_count); fc.overlap_max = uni(mid->overlap_max); fc.stereo_saving = uni(mid->stereo_saving);
    fc.intensity = uni(mid->intensity); fc.spec_avg = uni(mid->spec_avg);
    fc.stereo_narrow = uni(mid->pad[0]);
    const int max_data_bytes = uni(mid->max_data_bytes);
    int nbCompressedBytes = uni(mid->nbCompressedBytes), nbAvailableBytes = uni(mid->nbAvailableBytes);
    const i32 vbr_rate = uni(mid->vbr_rate), equiv_rate = uni(mid->equiv_rate);
    int effectiveBytes = uni(mid->effectiveBytes);
    i32 total_bits = uni(mid->total_bits);
    const int silence = uni(mid->silence), pitch_index = uni(mid->pitch_index), pf_on = uni(mid->pf_on);
    const i32 gain1 = uni(mid->gain1);
    const int prefilter_tapset = uni(mid->prefilter_tapset);
    const int isTransient = uni(mid->isTransient), shortBlocks = uni(mid->shortBlocks), tf_chan = uni(mid->tf_chan);
    const i32 tf_estimate = uni(mid->tf_estimate), temporal_vbr = uni(mid->temporal_vbr);
    const int transient_got_disabled = uni(mid->transient_got_disabled);
    RangeEnc enc;
#if defined(CA_LANE_FRAME)
    F.x16 = (x16_t *)const_cast<i16 *>(mid->X);
    F.packet = out;
    F.mid = const_cast<FrameMid *>(mid);
#endif
    enc.buf = F.packet + 1;
    enc.storage = uni(mid->ec_storage); enc.end_offs = uni(mid->ec_end_offs); enc.end_window = uni(mid->ec_end_window);
    enc.offs = uni(mid->ec_offs); enc.rng = uni(mid->ec_rng); enc.val = uni(mid->ec_val); enc.ext = uni(mid->ec_ext);
    enc.nend_bits = uni(mid->ec_nend_bits); enc.nbits_total = uni(mid->ec_nbits_total); enc.rem = uni(mid->ec_rem);
    enc.error = uni(mid->ec_error);
    {
#if !defined(CA_LANE_FRAME)
        i16 *X = frame_X(F);
        for (int k = lane(); k < 2 * FRAME; k += LANES) X[k] = mid->X[k];
#endif
#if !defined(CA_LANE_FRAME)
        for (int k = lane(); k < 2 * NB; k += LANES) {
            F.bandE[k] = mid->bandE[k];
            F.bandLogE[k] = mid->bandLogE[k];
            F.bandLogE2[k] = mid->bandLogE2[k];
            F.oldBandE[k] = mid->oldBandE[k];
            F.oldLogE[k] = mid->oldLogE[k];
            F.oldLogE2[k] = mid->oldLogE2[k];
        }
#endif
#if defined(CA_LANE_FRAME)
        {   // (two 16-byte loads in flight, not 32 byte loads each waited for)
            const v4i h0 = reinterpret_cast<const v4i *>(mid->packet_head)[0], h1 = reinterpret_cast<const v4i *>(mid->packet_head)[1];
            const u32 w[8] = { (u32)h0.x, (u32)h0.y, (u32)h0.z, (u32)h0.w, (u32)h1.x, (u32)h1.y, (u32)h1.z, (u32)h1.w };
#pragma unroll
            for (int k = 0; k < 32; k++) F.packet[1 + k] = (u8)(w[k >> 2] >> (8 * (k & 3)));
        }
#else
        for (int k = lane(); k < 32; k += LANES) F.packet[1 + k] = mid->packet_head[k];
#endif
    }
    i32 tell;
    wave_sync();
    // 11. TF resolution
    int tf_select;
    if (effectiveBytes >= 15 * C && cfg.complexity >= 2) {
        int lambda;
        if (effectiveBytes < 40) lambda = 12;
        else if (effectiveBytes < 60) lambda = 6;
        else if (effectiveBytes < 100) lambda = 4;
        else lambda = 3;
        lambda *= 2;
#if defined(CA_LANE_FRAME)
        tf_select = tf_analysis_lane(F, isTransient, lambda, tf_estimate, tf_chan);
    } else {
        F.tf_bits = isTransient ? (1u << NB) - 1 : 0u;
        tf_select = 0;
    }
    // the column is free from here to the PVQ walk: the energies move in (celt_enc_lane.h, slot map)
    {
        const Col loge = mcol(F, M_LOGE), olde = mcol(F, M_OLDE);
        CA_UNROLL_LANE
        for (int k = 0; k < 2 * NB; k++) { loge[k] = mid->bandLogE[k]; olde[k] = mid->oldBandE[k]; }
    }
#else
        tf_select = tf_analysis_wave(F, isTransient, lambda, tf_estimate, tf_chan);
    } else {
        for (int i = lane(); i < end; i += LANES) F.tf_res[i] = isTransient;
        wave_sync();
        tf_select = 0;
    }
#endif

    CA_TRACE("tf done tf_select=%d", tf_select); CA_TRACE("");
    CA_STAMP(7);
    // 12. coarse energy
#if defined(CA_LANE_FRAME)
    quant_coarse_energy_lane(F, fc, enc, (u32)total_bits, nbAvailableBytes, cfg.complexity >= 4, cfg.loss_rate);
#else
    quant_coarse_energy_wave(F, fc, enc, (u32)total_bits, nbAvailableBytes, cfg.complexity >= 4, cfg.loss_rate);
#endif

    CA_TRACE("coarse done tell=%d", ec_tell(enc)); CA_TRACE("");
#if !defined(CA_LANE_FRAME)
    CA_TAP("oldBandE_after_coarse", F.oldBandE, sizeof(F.oldBandE)); CA_TAP("error", F.error, sizeof(F.error));
#endif
    CA_STAMP(8);
    // 13. tf_encode, spread, dynalloc, trim
#if defined(CA_LANE_FRAME)
    tf_encode_lane(F, enc, isTransient, tf_select);
#else
    tf_encode_wave(F, enc, isTransient, tf_select);
#endif
    if (ec_tell(enc) + 4 <= total_bits) {
        if (shortBlocks || cfg.complexity < 3 || nbAvailableBytes < 10 * C) {
            fc.spread_decision = cfg.complexity == 0 ? SPREAD_NONE : SPREAD_NORMAL;
        } else {
            fc.spread_decision = spreading_decision_wave(F, fc, pf_on && !shortBlocks);
        }
        ec_enc_icdf(enc, fc.spread_decision, CLT_spread_icdf, 5);
    }
    i32 tot_boost;
    // opus_encode() hands int16 input over with lsb_depth 16; opus_encode_native takes the min with the ctl value
    // (src/opus_encoder.c:2022, :1034)
#if defined(CA_LANE_FRAME)
    const i32 maxDepth = dynalloc_analysis_lane(F, imin(16, cfg.lsb_depth), isTransient, celt_vbr, constrained_vbr, effectiveBytes, &tot_boost);
#else
    const i32 maxDepth = dynalloc_analysis_wave(F, fc, imin(16, cfg.lsb_depth), isTransient, celt_vbr, constrained_vbr,
                                                effectiveBytes, &tot_boost);
    for (int i = lane(); i < NB; i += LANES) {                                     // init_caps (celt.c:246-256)
        int Nb = (CLT_eband5ms[i + 1] - CLT_eband5ms[i]) << LM;
        F.cap[i] = ((CLT_cache_caps50[NB * (2 * LM + C - 1) + i] + 64) * C * Nb) >> 2;
    }
    wave_sync();
#endif
    int dynalloc_logp = 6;
    total_bits <<= BITRES;
    i32 total_boost = 0;
    tell = (i32)ec_tell_frac(enc);
    for (int i = 0; i < end; i++) {
        int width = (C * (CLT_eband5ms[i + 1] - CLT_eband5ms[i])) << LM;
        int quanta = imin(width << BITRES, imax(6 << BITRES, width));
        int dynalloc_loop_logp = dynalloc_logp;
        int boost = 0, j;
#if defined(CA_LANE_FRAME)
        const int off_i = mcol(F, M_OFFSETS)[i], cap_i = alloc_cap(i);
#else
        const int off_i = uni(F.offsets[i]), cap_i = uni(F.cap[i]);
#endif
        for (j = 0; tell + (dynalloc_loop_logp << BITRES) < total_bits - total_boost && boost < cap_i; j++) {
            int flag = j < off_i;
            ec_enc_bit_logp(enc, flag, (u32)dynalloc_loop_logp);
            tell = (i32)ec_tell_frac(enc);
            if (!flag) break;
            boost += quanta;
            total_boost += quanta;
            dynalloc_loop_logp = 1;
        }
        if (j) dynalloc_logp = imax(2, dynalloc_logp - 1);
#if defined(CA_LANE_FRAME)
        mcol(F, M_OFFSETS)[i] = (i16)boost;                                        // < cap + quanta <= 28 424
#else
        st0(&F.offsets[i], boost);
#endif
    }
    wave_sync();
    int dual_stereo = 0;
    if (C == 2) {
        dual_stereo = stereo_analysis_wave(F);
        fc.intensity = hysteresis_decision((i16)(equiv_rate / 1000), CLT_intensity_thresholds, CLT_intensity_histeresis, 21, fc.intensity);
        fc.intensity = imin(end, imax(0, fc.intensity));
    }
    int alloc_trim = 5;
    if (tell + (6 << BITRES) <= total_bits - total_boost) {
        alloc_trim = alloc_trim_analysis_wave(F, fc, tf_estimate, fc.intensity);
        ec_enc_icdf(enc, alloc_trim, CLT_trim_icdf, 7);
        tell = (i32)ec_tell_frac(enc);
    }

    CA_TRACE("trim done alloc_trim=%d tell=%d", alloc_trim, tell); CA_TRACE("");
    CA_STAMP(9);
    // 14. VBR target (celt_encoder.c:2002-2087)
    if (vbr_rate > 0) {
        const int lm_diff = 3 - LM;
        nbCompressedBytes = imin(nbCompressedBytes, 1275 >> (3 - LM));
        i32 base_target = vbr_rate - ((40 * C + 20) << BITRES);
        if (constrained_vbr) base_target += (fc.vbr_offset >> lm_diff);
        i32 target = compute_vbr_wave(fc, base_target, equiv_rate, constrained_vbr, tot_boost, tf_estimate, maxDepth, temporal_vbr);
        target = target + tell;
        i32 min_allowed = ((tell + total_boost + (1 << (BITRES + 3)) - 1) >> (BITRES + 3)) + 2 - nbFilledBytes;
        nbAvailableBytes = (target + (1 << (BITRES + 2))) >> (BITRES + 3);
        nbAvailableBytes = imax(min_allowed, nbAvailableBytes);
        nbAvailableBytes = imin(nbCompressedBytes, nbAvailableBytes + nbFilledBytes) - nbFilledBytes;
        i32 delta = target - vbr_rate;
        target = nbAvailableBytes << (BITRES + 3);
        if (silence) {
            nbAvailableBytes = 2;
            target = (2 * 8) << BITRES;
            delta = 0;
        }
        i32 alpha;
        if (fc.vbr_count < 970) {
            fc.vbr_count++;
            alpha = (i16)celt_rcp(shl32(fc.vbr_count + 20, 16));
        } else {
            alpha = 33;                                                             // QCONST16(.001f,15)
        }
        if (constrained_vbr) fc.vbr_reservoir += target - vbr_rate;
        if (constrained_vbr) {
            fc.vbr_drift += mul16_32_q15(alpha, (delta * (1 << lm_diff)) - fc.vbr_offset - fc.vbr_drift);
            fc.vbr_offset = -fc.vbr_drift;
        }
        if (constrained_vbr && fc.vbr_reservoir < 0) {
            int adjust = (-fc.vbr_reservoir) / (8 << BITRES);
            nbAvailableBytes += silence ? 0 : adjust;
            fc.vbr_reservoir = 0;
        }
        nbCompressedBytes = imin(nbCompressedBytes, nbAvailableBytes + nbFilledBytes);
        ec_enc_shrink(enc, (u32)nbCompressedBytes);
        wave_sync();
    }

    CA_TRACE("vbr done nbCompressedBytes=%d", nbCompressedBytes); CA_TRACE("");
    CA_STAMP(10);
    // 15. allocation
    i32 bits = (((i32)nbCompressedBytes * 8) << BITRES) - (i32)ec_tell_frac(enc) - 1;
    const int anti_collapse_rsv = isTransient && LM >= 2 && bits >= ((LM + 2) << BITRES) ? (1 << BITRES) : 0;
    bits -= anti_collapse_rsv;
    const int signalBandwidth = end - 1;
#if defined(CA_LANE_FRAME)
    AllocOut ao = compute_allocation_lane(F, enc, alloc_trim, fc.intensity, dual_stereo, bits, fc.lastCodedBands, signalBandwidth);
#else
    AllocOut ao = compute_allocation_wave(F, enc, C, alloc_trim, fc.intensity, dual_stereo, bits, fc.lastCodedBands, signalBandwidth);
#endif
    fc.intensity = ao.intensity;
    dual_stereo = ao.dual_stereo;
    const int codedBands = ao.codedBands;
    if (fc.lastCodedBands) fc.lastCodedBands = imin(fc.lastCodedBands + 1, imax(fc.lastCodedBands - 1, codedBands));
    else fc.lastCodedBands = codedBands;

    CA_TRACE("alloc done codedBands=%d", codedBands); CA_TRACE("");
#if !defined(CA_LANE_FRAME)
    CA_TAP("pulses", F.pulses, sizeof(F.pulses)); CA_TAP("fine_quant", F.fine_quant, sizeof(F.fine_quant)); CA_TAP("tf_res", F.tf_res, sizeof(F.tf_res));
#endif
    CA_STAMP(11);
    // 16. fine energy
#if defined(CA_LANE_FRAME)
    quant_fine_energy_lane(F, enc);
    park_energy_state_lane(F);                     // the PVQ walk takes slots 0..239 of the column
#else
    quant_fine_energy_wave(F, enc, C);
#endif

    CA_STAMP(12);
    // 17. PVQ
    quant_all_bands_wave(F, enc, C, shortBlocks, fc.spread_decision, dual_stereo, fc.intensity,
                         nbCompressedBytes * (8 << BITRES) - anti_collapse_rsv, ao.balance, codedBands);

    CA_TRACE("pvq done tell=%d", ec_tell(enc)); CA_TRACE("");
    CA_STAMP(13);
    // 18. anti-collapse bit, energy finalise, state roll-over
    if (anti_collapse_rsv > 0) {
        int anti_collapse_on = fc.consec_transient < 2;
        ec_enc_bits(enc, (u32)anti_collapse_on, 1);
    }
#if defined(CA_LANE_FRAME)
    quant_energy_finalise_lane(F, enc, nbCompressedBytes * 8 - ec_tell(enc));
    fc.prefilter_period = pitch_index;
    fc.prefilter_gain = gain1;
    fc.prefilter_tapset = prefilter_tapset;
    if (st_out) {
        // energy histories of the stream (celt_encoder.c:2215-2238), straight from the hand-off record to the state
        for (int k = 0; k < C * NB; k++) {
            const i16 ob = silence ? (i16)-28672 : mid->oldBandE[k];
            const i16 ol = mid->oldLogE[k];
            st_out->oldBandE[k] = ob;
            if (!isTransient) { st_out->oldLogE2[k] = ol; st_out->oldLogE[k] = ob; }
            else { st_out->oldLogE2[k] = mid->oldLogE2[k]; st_out->oldLogE[k] = (i16)imin(ol, ob); }
        }
    }
#else
    quant_energy_finalise_wave(F, enc, nbCompressedBytes * 8 - ec_tell(enc), C);
    if (silence)
        for (int k = lane(); k < C * NB; k += LANES) F.oldBandE[k] = -28672;
    fc.prefilter_period = pitch_index;
    fc.prefilter_gain = gain1;
    fc.prefilter_tapset = prefilter_tapset;
    wave_sync();
    if (!isTransient) {
        for (int k = lane(); k < C * NB; k += LANES) { F.oldLogE2[k] = F.oldLogE[k]; F.oldLogE[k] = F.oldBandE[k]; }
    } else {
        for (int k = lane(); k < C * NB; k += LANES) F.oldLogE[k] = (i16)imin(F.oldLogE[k], F.oldBandE[k]);
    }
#endif
    if (isTransient || transient_got_disabled) fc.consec_transient++;
    else fc.consec_transient = 0;
    fc.rng = enc.rng;

    CA_STAMP(14);
    // 19. flush
    ec_enc_done(enc);

    // Opus layer: TOC (gen_toc: CELT-only, 20 ms, fullband) + length (opus_encoder.c:1925-1970)
    st0(&F.packet[0], (u8)(0x80 | (3 << 5) | (3 << 3) | ((C == 2) << 2)));
    int ret = nbCompressedBytes + 1;
    wave_sync();
    if (!cfg.vbr) {
        // opus_packet_pad to max_data_bytes: a CELT CBR packet already has that size (ret == max_data_bytes)
        if (ret != max_data_bytes) enc.error = -1;
    }
#if defined(CA_LANE_FRAME)
    if (st_out) store_state_scalars(fc, st_out);
#else
    if (st_out) store_state(fc, F, st_out);
#endif
    if (lane() == 0)
        for (int k = ret; k < ((ret + 3) & ~3); k++) F.packet[k] = 0;               // deterministic pad bytes
    wave_sync();
#if !defined(CA_LANE_FRAME)
    {   // packet -> HBM (word-wise; the slab stride is a multiple of 4)
        const u32 *src = reinterpret_cast<const u32 *>(F.packet);
        u32 *dst = reinterpret_cast<u32 *>(out);
        for (int k = lane(); k < (ret + 3) / 4; k += LANES) dst[k] = src[k];
    }
#endif
    CA_STAMP(15);
    FrameResult r;
    r.bytes = enc.error ? -3 : ret;                                                  // OPUS_INTERNAL_ERROR
    r.final_range = fc.rng;
    return r;
}

}  // namespace ca

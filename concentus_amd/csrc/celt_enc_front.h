// celt_enc_front.h -- front half of the CELT frame encoder: PCM -> normalised MDCT bands + analysis.
//
// Wave-cooperative (one wavefront per frame) counterparts of, in call order:
//   dc_reject                  opus-fix/src/opus_encoder.c:362-384
//   celt_preemphasis           celt/celt_encoder.c:464-548 (fast path :476-488)
//   run_prefilter              celt/celt_encoder.c:1067-1193
//     pitch_downsample / celt_fir5 / _celt_autocorr / _celt_lpc     celt/pitch.c:147-223,105-145; celt/celt_lpc.c:232-330,37-92
//     pitch_search / find_best_pitch / celt_pitch_xcorr              celt/pitch.c:260-369,45-103,225-258
//     remove_doubling                                                celt/pitch.c:372-503
//     comb_filter                                                    celt/celt.c:183-237
//   transient_analysis         celt/celt_encoder.c:227-377
//   compute_band_energies, normalise_bands, amp2Log2                 celt/bands.c:97-168, celt/quant_bands.c:551-575
// Data-parallel loops are strided over the lanes; scalar decisions run identically on every lane and
// only lane 0 stores ("st0"). The arithmetic of every element follows the reference bit for bit.
#pragma once
#include "celt_math.h"
#include "celt_state.h"
#include "device_tables.h"
#include "mdct_dev.h"
#include "rangecoder.h"

namespace ca {

// Everything is inlined into the frame kernel: a non-inlined device call on gfx9 saves/restores the live
// VGPRs through scratch and forces by-reference state (the range coder) into memory -- measured as
// ~60 GB of scratch traffic per 65 536-frame launch in the first version (profiles/r01_celt_first).
#if defined(CA_HOST_EMU)
#define CA_DEVFN static inline
#else
#define CA_DEVFN __device__ __forceinline__
#endif

enum { NB = 21, OVL = 120, FRAME = 960, MAXP = 1024, MINP = 15, LM3 = 3, M8 = 8 };
enum { SPREAD_NONE = 0, SPREAD_LIGHT = 1, SPREAD_NORMAL = 2, SPREAD_AGGRESSIVE = 3 };

template <class P, class V> CA_DEV void st0(P p, V v) { if (lane() == 0) *p = v; }

// ---- per-frame LDS working sets (one per wavefront) --------------------------------------------------
// The frame is encoded by two kernels so that each phase keeps only what it needs in LDS:
//   front (PCM -> normalised bands X + band energies + the first coded flags): the time-domain buffers
//         dominate (~23 KB per wave);
//   back  (TF analysis ... PVQ ... range-coder flush): X, a small scratch and the packet (~9.5 KB per
//         wave), which lets 16 waves share a CU and overlap their serial, latency-bound chains.
// The hand-off between them is the pointer-free FrameMid record in HBM (celt_enc.h).
struct __attribute__((aligned(16))) FrontLds {
    enum { IN_IS_GLOBAL = 0, XF_CHANNELS = 2 };
    i32 in[2][FRAME + OVL];        // overlap + pre-emphasised (then comb-filtered) signal; finally X[2][960] (i16)
    i32 xf[2][FRAME];              // dc-rejected PCM (i16) -> unfiltered pre-emphasised samples -> MDCT coefficients
    union {
        i16 raw_pcm[2 * FRAME];                                     // interleaved input
        struct { i16 buf[992]; i16 xlp4[240]; i16 ylp4[484]; i32 xcorr[520]; } pitch;
        i16 trans[2][FRAME + OVL];
        int2 f2[480];
    } s;
    u8 packet[64];                 // only the first few range-coder bytes are produced in this phase
    i32 bandE[2 * NB];
    i16 bandLogE[2 * NB], bandLogE2[2 * NB];
    i16 oldBandE[2 * NB], oldLogE[2 * NB], oldLogE2[2 * NB];
    i16 follower[2 * NB];
    i16 normg[2 * NB];
    i8 normshift[2 * NB];
    i32 scal[16];                  // scalar hand-off slots (lane-local results published to the wave)
    void *diag;                    // StageClock* in the diagnostic build, unused otherwise
};

#if defined(CA_LANE_FRAME)
// Lane-per-frame build: NO per-frame arrays in private memory (celt_enc_lane.h). What a lane owns is a column of the
// workgroup's LDS scratch ([slot][lane], LS_SLOTS 16-bit slots), its rows in HBM, and a handful of scalars; this struct only
// carries the pointers (their TYPES carry the address space, wave.h) and the scalars, all of which live in registers.
enum { LS_SLOTS = 264, LS_PULSES = 240 };      // slots 0..239: band buffers of the PVQ walk / staging of the stages before it
struct FrameMid;
struct BackLds {
    x16_t *x16;                    // -> FrameMid::X of this frame (consumed band by band), HBM
    u8 *packet;                    // -> the output slab of this frame
    CA_AS_LDS i16 *col;            // -> slot 0 of this lane's column
    FrameMid *mid;                 // -> the frame's hand-off record: inputs, and parking space for what must survive the PVQ walk
    void *diag;                    // StageClock* in the diagnostic build
    u32 tf_bits;                   // tf_res decision per band (bit i); tf_change(i) = tf_select_table[LM][tf_sel + bit]
    int tf_sel;
    // parked second children of split partitions (quant_band_lane): three packed words each, selected by explicit compares --
    // a run-time index would put them in memory
    u32 ps0[4], ps1[4], ps2[4];
};
#else
struct __attribute__((aligned(16))) BackLds {
    i16 x16[2 * FRAME];            // normalised bands X[c*960 + j]
    union {
        struct { i16 tmp[176]; i16 tmp1[176]; } tf;
        struct { i16 y[176]; i32 iy[176]; i16 xabs[176]; } pvq;
    } s;
    u8 packet[1280];
    i32 bandE[2 * NB];
    i16 bandLogE[2 * NB], bandLogE2[2 * NB], error[2 * NB];
    i16 oldBandE[2 * NB], oldLogE[2 * NB], oldLogE2[2 * NB];
    i16 oldE_intra[2 * NB], error_intra[2 * NB];
    i16 follower[2 * NB], noise_floor[NB];
    u8 coarse_save[256];
    i32 offsets[NB], cap[NB], pulses[NB], fine_quant[NB], fine_priority[NB], tf_res[NB];
    i32 bits1[NB], bits2[NB], thresh[NB], trim_offset[NB];
    i32 metric[NB], path0[NB], path1[NB];
    i32 scal[16];
    void *diag;
    i32 pstack[4][8];              // parked second children of split partitions (quant_band_wave)
};
#endif

// Working sets of the two halves of the split front phase. The [2][1080] time signal is not staged in LDS
// there: phase 1 produces it (comb filter output) and phase 2 consumes it (MDCT fold) exactly once, so both
// go straight to the HBM hand-off buffer, and the normalised bands X are written straight into FrameMid.
// 14.4 KB / 12.7 KB instead of 22.8 KB -> 11 / 12 waves per CU instead of 7.
struct __attribute__((aligned(16))) Front1Lds {
    enum { IN_IS_GLOBAL = 1, XF_CHANNELS = 2 };
    i32 *in_g;                     // -> in_ws[2][1080] of this frame
    i32 xf[2][FRAME];
    union {
        struct { i16 buf[992]; i16 xlp4[240]; i16 ylp4[484]; i32 xcorr[520]; } pitch;
    } s;
    u8 packet[64];
    i32 bandE[2 * NB];
    i16 bandLogE[2 * NB], bandLogE2[2 * NB];
    i16 oldBandE[2 * NB], oldLogE[2 * NB], oldLogE2[2 * NB];
    i32 scal[16];
    void *diag;
};

struct __attribute__((aligned(16))) Front2Lds {
    // XF_CHANNELS 1 (round 3): ONE channel's MDCT coefficients at a time. Transform, band energies and normalisation of a
    // channel need nothing of the other one, so the channels go through the same buffer one after the other and the
    // normalised bands leave for FrameMid::X at once (celt_encode_front_phase). 8.7 KB instead of 12.5 KB of LDS per
    // wavefront: four wavefronts per SIMD (the register budget's limit) instead of three.
    enum { IN_IS_GLOBAL = 1, XF_CHANNELS = 1 };
    i32 *in_g;                     // -> in_ws[2][1080] of this frame (read only here)
    i16 *x_g;                      // -> FrameMid::X of this frame
    i32 xf[1][FRAME];
    union {
        int2 f2[480];
    } s;
    u8 packet[64];
    i32 bandE[2 * NB];
    i16 bandLogE[2 * NB], bandLogE2[2 * NB];
    i16 oldBandE[2 * NB], oldLogE[2 * NB], oldLogE2[2 * NB];
    i16 follower[2 * NB];
    i16 normg[2 * NB];
    i8 normshift[2 * NB];
    i32 scal[16];
    void *diag;
};

// time signal of channel c ([1080]: 120 overlap + 960 new samples)
CA_DEV i32 *tsig(FrontLds &F, int c) { return F.in[c]; }
CA_DEV i32 *tsig(Front1Lds &F, int c) { return F.in_g + c * (FRAME + OVL); }
CA_DEV i32 *tsig(Front2Lds &F, int c) { return F.in_g + c * (FRAME + OVL); }

CA_DEV i16 *frame_X(FrontLds &F) { return reinterpret_cast<i16 *>(&F.in[0][0]); }      // X[c*960 + j]
CA_DEV i16 *frame_X(Front2Lds &F) { return F.x_g; }
CA_DEV i16 *frame_pcmf(Front1Lds &F) { return reinterpret_cast<i16 *>(&F.xf[0][0]); }
CA_DEV x16_t *frame_X(BackLds &F) { return F.x16; }
CA_DEV i16 *frame_pcmf(FrontLds &F) { return reinterpret_cast<i16 *>(&F.xf[0][0]); }   // pcmf[c*960 + i]

// Uniform per-frame scalars (identical in every lane).
struct FrameCtx {
    int C;
    // stream state (celt_encoder.c:82-120)
    i32 hp_mem[4];
    u32 rng;
    int spread_decision, tonal_average, lastCodedBands, hf_average, tapset_decision;
    i32 delayedIntra;
    int prefilter_period, prefilter_gain, prefilter_tapset, consec_transient;
    i32 preemph_memE[2];
    i32 vbr_reservoir, vbr_drift, vbr_offset, vbr_count, overlap_max;
    int stereo_saving, intensity, spec_avg;
    int stereo_narrow;             // 16384 - hybrid_stereo_width_Q14 of the Opus layer (src/opus_encoder.c:91): 0 = full width
    const i32 *hist;               // prefilter_mem of the stream in HBM, or nullptr (all zero)
};

// ---- dc_reject (src/opus_encoder.c:362-384) + celt_maxabs16 ------------------------------------------
// Serial per channel: lane c filters channel c. cutoff 3 Hz @ 48 kHz -> shift = ilog2(48000/9) = 12.
template <class L>
CA_DEV void dc_reject_wave(L &F, FrameCtx &fc)
{
    const int C = fc.C;
    i16 *pcmf = frame_pcmf(F);
    for (int c = lane(); c < C; c += LANES) {
        i32 m0 = fc.hp_mem[2 * c], m1 = fc.hp_mem[2 * c + 1];
        for (int i = 0; i < FRAME; i++) {
            i32 x = shl32(F.s.raw_pcm[C * i + c], 15);
            i32 tmp = sub32(x, m0);
            m0 = add32(m0, pshr32(sub32(x, m0), 12));
            i32 y = sub32(tmp, m1);
            m1 = add32(m1, pshr32(sub32(tmp, m1), 12));
            i32 o = pshr32(y, 15);
            o = o > 32767 ? 32767 : (o < -32767 ? -32767 : o);          // SATURATE(x, 32767)
            pcmf[c * FRAME + i] = (i16)o;
        }
        F.scal[2 * c] = m0;
        F.scal[2 * c + 1] = m1;
    }
    wave_sync();
    for (int c = 0; c < C; c++) { fc.hp_mem[2 * c] = F.scal[2 * c]; fc.hp_mem[2 * c + 1] = F.scal[2 * c + 1]; }
    wave_sync();
}

// Stereo width reduction of the Opus layer at low rates (src/opus_encoder.c:1790-1809 + stereo_fade :411-441):
// below 38.2 kb/s the side signal is attenuated, cross-fading from the previous frame's width over the overlap.
// Works on the planar dc-rejected PCM; equiv_rate == bitrate at 50 frames/s.
template <class L>
CA_DEV void stereo_width_wave(L &F, FrameCtx &fc, i32 bitrate_bps)
{
    const i32 width = imin(1 << 14, 2 * imax(0, bitrate_bps - 30000)), prev = (1 << 14) - fc.stereo_narrow;
    if (!(prev < (1 << 14) || width < (1 << 14))) return;
    i32 g1 = prev == 16384 ? 32767 : shl16(prev, 1), g2 = width == 16384 ? 32767 : shl16(width, 1);
    g1 = (i16)(32767 - g1);
    g2 = (i16)(32767 - g2);
    i16 *pcmf = frame_pcmf(F);
    for (int i = lane(); i < FRAME; i += LANES) {
        i32 g = g2;
        if (i < OVL) {
            i32 w = (i16)mul16_16_q15(CLT_window120[i], CLT_window120[i]);
            g = (i16)(mac16_16(mul16_16(w, g2), (i16)(32767 - w), g1) >> 15);
        }
        i32 l = pcmf[i], r = pcmf[FRAME + i];
        i32 diff = (i16)((l - r) >> 1);
        diff = mul16_16_q15(g, diff);
        pcmf[i] = (i16)(l - diff);
        pcmf[FRAME + i] = (i16)(r + diff);
    }
    fc.stereo_narrow = (1 << 14) - width;
    wave_sync();
}

// celt_maxabs16 over samples [i0, i1) of all channels (mathops.h:47-58)
template <class L>
CA_DEV i32 maxabs_pcm(L &F, int C, int i0, int i1)
{
    const i16 *pcmf = frame_pcmf(F);
    i32 mx = 0, mn = 0;
    for (int c = 0; c < C; c++)
        for (int i = i0 + lane(); i < i1; i += LANES) {
            i32 v = pcmf[c * FRAME + i];
            mx = imax(mx, v);
            mn = imin(mn, v);
        }
    mx = wave_max(mx);
    mn = wave_min(mn);
    return imax(mx, -mn);
}

// ---- celt_preemphasis, fast path (celt_encoder.c:476-488): coef0 = 27853, SIG_SHIFT = 12 -----------
template <class L>
CA_DEV void preemphasis_wave(L &F, FrameCtx &fc)
{
    const i16 *pcmf = frame_pcmf(F);
    if (L::IN_IS_GLOBAL) {
        // split pipeline: the pre-emphasised samples are only needed in xf (the pitch analysis' and the comb
        // filter's input), so they are written there directly instead of through the HBM time-signal buffer.
        // pcmf aliases the first half of xf: channel 1 goes first (its xf half is clear of pcmf), channel 0
        // walks down in blocks whose reads complete before their writes (xf[0][i] lands on pcmf[2i], pcmf[2i+1]).
        for (int c = fc.C - 1; c >= 0; c--) {
            // (no fc.preemph_memE[c]: a run-time index into FrameCtx would send the whole struct to scratch memory)
            const i32 mem0 = c ? fc.preemph_memE[1] : fc.preemph_memE[0];
            const i32 memn = mul16_16(27853, pcmf[c * FRAME + FRAME - 1]) >> 3;
            if (c) fc.preemph_memE[1] = memn; else fc.preemph_memE[0] = memn;
            wave_sync();
            for (int base = FRAME - LANES; base >= 0; base -= LANES) {
                const int i = base + lane();
                i32 x = pcmf[c * FRAME + i];
                i32 m = i == 0 ? mem0 : (mul16_16(27853, pcmf[c * FRAME + i - 1]) >> 3);
                wave_sync();
                F.xf[c][i] = sub32(shl32(x, 12), m);
                wave_sync();
            }
        }
        return;
    }
    for (int c = 0; c < fc.C; c++) {
        for (int i = lane(); i < FRAME; i += LANES) {
            i32 x = pcmf[c * FRAME + i];
            i32 m = i == 0 ? (c ? fc.preemph_memE[1] : fc.preemph_memE[0]) : (mul16_16(27853, pcmf[c * FRAME + i - 1]) >> 3);
            tsig(F, c)[OVL + i] = sub32(shl32(x, 12), m);
        }
        const i32 memn = mul16_16(27853, pcmf[c * FRAME + FRAME - 1]) >> 3;
        if (c) fc.preemph_memE[1] = memn; else fc.preemph_memE[0] = memn;
    }
    wave_sync();
}

// ---- pitch analysis --------------------------------------------------------------------------------
// pre[c][k], k in [0, 1984): 1024 samples of history (stream state, HBM) followed by the 960 new ones.
template <class L>
CA_DEV i32 pre_at(const L &F, const FrameCtx &fc, int c, int k)
{
    if (k >= MAXP) return F.xf[c][k - MAXP];
    return fc.hist ? fc.hist[c * MAXP + k] : 0;
}

template <class L>
CA_DEVFN void pitch_downsample_wave(L &F, const FrameCtx &fc)              // pitch.c:147-223
{
    const int C = fc.C, len = MAXP + FRAME;
    i16 *x_lp = F.s.pitch.buf;
    // The first frame of a stream (every frame of the independent-frames workload) has an all-zero history: pre[c][k] = 0
    // for k < 1024. Zeros change neither the extrema nor any of the wrapping sums below, and they decimate / filter to
    // zeros, so those ranges are skipped (x_lp[0..512) is simply cleared).
    const int k0 = fc.hist ? 0 : MAXP, h0 = k0 >> 1;
    i32 mx = 0, mn = 0;                                                            // celt_maxabs32 per channel
    for (int c = 0; c < C; c++)
        for (int k = k0 + lane(); k < len; k += LANES) {
            i32 v = pre_at(F, fc, c, k);
            mx = imax(mx, v);
            mn = imin(mn, v);
        }
    i32 maxabs = imax(wave_max(mx), neg32(wave_min(mn)));
    if (maxabs < 1) maxabs = 1;
    int shift = celt_ilog2(maxabs) - 10;
    if (shift < 0) shift = 0;
    if (C == 2) shift++;
    for (int i = lane(); i < h0; i += LANES) x_lp[i] = 0;
    for (int i = h0 + lane(); i < (len >> 1); i += LANES) {
        i32 acc = 0;
        for (int c = 0; c < C; c++) {
            i32 v;
            if (i == 0)
                v = (add32(pre_at(F, fc, c, 1) >> 1, pre_at(F, fc, c, 0)) >> 1) >> shift;
            else
                v = (add32(add32(pre_at(F, fc, c, 2 * i - 1), pre_at(F, fc, c, 2 * i + 1)) >> 1,
                           pre_at(F, fc, c, 2 * i)) >> 1) >> shift;
            acc = c == 0 ? (i16)v : (i16)(acc + v);                               // x_lp is opus_val16
        }
        x_lp[i] = (i16)acc;
    }
    wave_sync();

    // _celt_autocorr(x_lp, ac, NULL, 0, 4, n = 992)  (celt_lpc.c:232-330), overlap == 0
    const int n = len >> 1, lag = 4, fastN = n - lag;
    i32 part = 0;
    for (int i = h0 + lane(); i < n; i += LANES) part = add32(part, mul16_16(x_lp[i], x_lp[i]) >> 9);
    i32 ac0 = add32(1 + (n << 7), wave_add(part));
    int sh = (celt_ilog2(ac0) - 30 + 10) / 2;
    i32 ac[5];
    if (sh <= 0) sh = 0;
    // ac[k] = sum_{i=0}^{n-1-k} xs[i]*xs[i+k] with xs = PSHR32(x_lp, sh): the reference's fastN split
    // (celt_pitch_xcorr over fastN + tail loop) adds up to exactly this, and MAC16_16 sums wrap.
    (void)fastN;
    for (int k = 0; k <= lag; k++) {
        i32 p = 0;
        for (int i = (h0 ? h0 - LANES : 0) + lane(); i + k < n; i += LANES) {       // products with x_lp[i < h0] are zero
            i32 a = sh ? (i16)pshr32(x_lp[i], sh) : x_lp[i];
            i32 b = sh ? (i16)pshr32(x_lp[i + k], sh) : x_lp[i + k];
            p = mac16_16(p, a, b);
        }
        ac[k] = wave_add(p);
    }
    int shift2x = 2 * sh;
    if (shift2x <= 0) ac[0] = add32(ac[0], shl32(1, -shift2x));
    if (ac[0] < 268435456) {
        int s2 = 29 - ec_ilog((u32)ac[0]);
        for (int k = 0; k <= lag; k++) ac[k] = shl32(ac[k], s2);
    } else if (ac[0] >= 536870912) {
        int s2 = 1;
        if (ac[0] >= 1073741824) s2++;
        for (int k = 0; k <= lag; k++) ac[k] = ac[k] >> s2;
    }
    // noise floor -40 dB, lag windowing (pitch.c:186-199)
    ac[0] = add32(ac[0], ac[0] >> 13);
    for (int i = 1; i <= 4; i++) ac[i] = sub32(ac[i], mul16_32_q15(2 * i * i, ac[i]));
    // _celt_lpc(lpc, ac, 4)  (celt_lpc.c:37-92)
    i32 lpc32[4] = {0, 0, 0, 0};
    i32 err = ac[0];
    if (ac[0] != 0) {
        for (int i = 0; i < 4; i++) {
            i32 rr = 0;
            for (int j = 0; j < i; j++) rr = add32(rr, mul32_32_q31(lpc32[j], ac[i - j]));
            rr = add32(rr, ac[i + 1] >> 3);
            i32 r = neg32(frac_div32(shl32(rr, 3), err));
            lpc32[i] = r >> 3;
            for (int j = 0; j < ((i + 1) >> 1); j++) {
                i32 t1 = lpc32[j], t2 = lpc32[i - 1 - j];
                lpc32[j] = add32(t1, mul32_32_q31(r, t2));
                lpc32[i - 1 - j] = add32(t2, mul32_32_q31(r, t1));
            }
            err = sub32(err, mul32_32_q31(mul32_32_q31(r, r), err));
            if (err < (ac[0] >> 10)) break;
        }
    }
    i32 lpc[4], tmp = 32767;
    for (int i = 0; i < 4; i++) {
        i32 l = (i16)pshr32(lpc32[i], 16);                                        // ROUND16(lpc,16)
        tmp = (i16)mul16_16_q15(29491, tmp);                                      // QCONST16(.9f,15)
        lpc[i] = (i16)mul16_16_q15(l, tmp);
    }
    const i32 c1 = 26214;                                                          // QCONST16(.8f,15)
    i32 num[5];
    num[0] = (i16)(lpc[0] + 3277);                                                 // QCONST16(.8f,SIG_SHIFT)
    num[1] = (i16)(lpc[1] + mul16_16_q15(c1, lpc[0]));
    num[2] = (i16)(lpc[2] + mul16_16_q15(c1, lpc[1]));
    num[3] = (i16)(lpc[3] + mul16_16_q15(c1, lpc[2]));
    num[4] = (i16)mul16_16_q15(c1, lpc[3]);
    // celt_fir5 in place, zero initial memory (pitch.c:105-145). Chunks go from the end so that the taps
    // x[i-1..i-5] are still unfiltered when read.
    for (int base = ((n - 1) / LANES) * LANES; base >= h0; base -= LANES) {       // below h0: zeros in, zeros out
        int i = base + lane();
        i32 y = 0;
        if (i < n) {
            i32 sum = shl32(x_lp[i], 12);
            for (int k = 0; k < 5; k++) {
                i32 xm = i - 1 - k >= 0 ? x_lp[i - 1 - k] : 0;
                sum = mac16_16(sum, num[k], xm);
            }
            y = (i16)pshr32(sum, 12);
        }
        wave_sync();
        if (i < n) x_lp[i] = (i16)y;
        wave_sync();
    }
}

// find_best_pitch (pitch.c:45-103): inherently ordered (cross-multiplied comparisons against the running
// best two), so it runs as uniform scalar code on every lane.
CA_DEV void find_best_pitch_uniform(const i32 *xcorr, const i16 *y, int len, int max_pitch, int *best_pitch,
                                    int yshift, i32 maxcorr)
{
    i32 Syy = 1;
    i32 best_num[2] = {-1, -1};
    i32 best_den[2] = {0, 0};
    int xshift = celt_ilog2(maxcorr) - 14;
    best_pitch[0] = 0;
    best_pitch[1] = 1;
    {
        i32 p = 0;
        for (int j = lane(); j < len; j += LANES) p = add32(p, mul16_16(y[j], y[j]) >> yshift);
        Syy = add32(Syy, wave_add(p));
    }
    for (int i = 0; i < max_pitch; i++) {
        if (xcorr[i] > 0) {
            i32 xcorr16 = (i16)vshr32(xcorr[i], xshift);
            i32 num = (i16)mul16_16_q15(xcorr16, xcorr16);
            if (mul16_32_q15(num, best_den[1]) > mul16_32_q15(best_num[1], Syy)) {
                if (mul16_32_q15(num, best_den[0]) > mul16_32_q15(best_num[0], Syy)) {
                    best_num[1] = best_num[0];
                    best_den[1] = best_den[0];
                    best_pitch[1] = best_pitch[0];
                    best_num[0] = num;
                    best_den[0] = Syy;
                    best_pitch[0] = i;
                } else {
                    best_num[1] = num;
                    best_den[1] = Syy;
                    best_pitch[1] = i;
                }
            }
        }
        Syy = add32(Syy, sub32(mul16_16(y[i + len], y[i + len]) >> yshift, mul16_16(y[i], y[i]) >> yshift));
        Syy = imax(1, Syy);
    }
}

// find_best_pitch on the whole wave, exactly the sequential result. The running top-2 changes only at
// "record" lags, so each block of LANES lags is tested against the current state in parallel; the first lag
// that passes is applied, the lags after it are re-tested, and so on (a lag that fails against the state it
// would see sequentially never updates anything). Syy is a prefix sum as long as its max(1, .) clamp never
// triggers; that is checked first and the uniform scalar loop above is kept for the (silence-like) rest.
CA_DEV void find_best_pitch_wave(const i32 *xcorr, const i16 *y, int len, int max_pitch, int *best_pitch,
                                 int yshift, i32 maxcorr)
{
    i32 Syy0;
    {
        i32 p = 0;
        for (int j = lane(); j < len; j += LANES) p = add32(p, mul16_16(y[j], y[j]) >> yshift);
        Syy0 = add32(1, wave_add(p));
    }
    // pass 1: would the clamp ever trigger?
    i32 run = Syy0, mn = Syy0;
    for (int base = 0; base < max_pitch; base += LANES) {
        const int i = base + lane();
        i32 d = 0;
        if (i < max_pitch) d = sub32(mul16_16(y[i + len], y[i + len]) >> yshift, mul16_16(y[i], y[i]) >> yshift);
        i32 inc = wave_scan_add(d);
        mn = imin(mn, add32(run, inc));
        run = add32(run, wave_last(inc));
    }
    if (wave_min(mn) < 1) { find_best_pitch_uniform(xcorr, y, len, max_pitch, best_pitch, yshift, maxcorr); return; }
    i32 bn0 = -1, bn1 = -1, bd0 = 0, bd1 = 0;
    int bp0 = 0, bp1 = 1;
    const int xshift = celt_ilog2(maxcorr) - 14;
    run = Syy0;
    for (int base = 0; base < max_pitch; base += LANES) {
        const int i = base + lane();
        i32 d = 0, xc = 0;
        if (i < max_pitch) {
            d = sub32(mul16_16(y[i + len], y[i + len]) >> yshift, mul16_16(y[i], y[i]) >> yshift);
            xc = xcorr[i];
        }
        const i32 inc = wave_scan_add(d);
        const i32 S = add32(run, sub32(inc, d));              // Syy as lag i sees it
        run = add32(run, wave_last(inc));
        const i32 x16 = (i16)vshr32(xc, xshift);
        const i32 num = (i16)mul16_16_q15(x16, x16);
        bool pending = xc > 0;
        for (;;) {
            const bool pass = pending && mul16_32_q15(num, bd1) > mul16_32_q15(bn1, S);
            const uint64_t m = wave_ballot(pass);
            if (m == 0) break;
            const int l = __builtin_ctzll(m);
            const i32 ne = lane_bcast(num, l), Se = lane_bcast(S, l);
            if (mul16_32_q15(ne, bd0) > mul16_32_q15(bn0, Se)) {
                bn1 = bn0; bd1 = bd0; bp1 = bp0;
                bn0 = ne; bd0 = Se; bp0 = base + l;
            } else {
                bn1 = ne; bd1 = Se; bp1 = base + l;
            }
            pending = pending && lane() > l;
        }
    }
    best_pitch[0] = bp0;
    best_pitch[1] = bp1;
}

// pitch_search(x_lp = buf+512, y = buf, len = 960, max_pitch = 979)  (pitch.c:260-369)
template <class L>
CA_DEVFN int pitch_search_wave(L &F, bool zero_hist)
{
    const int len = FRAME, max_pitch = MAXP - 3 * MINP, lag = len + max_pitch;
    const i16 *y = F.s.pitch.buf, *x_lp = F.s.pitch.buf + (MAXP >> 1);
    i16 *x4 = F.s.pitch.xlp4, *y4 = F.s.pitch.ylp4;
    i32 *xcorr = F.s.pitch.xcorr;
    i32 mx = 0, mn = 0, my = 0, ny = 0;
    for (int j = lane(); j < (len >> 2); j += LANES) { i32 v = x_lp[2 * j]; x4[j] = (i16)v; mx = imax(mx, v); mn = imin(mn, v); }
    // first frame of a stream: y[0..512) is the all-zero history (pitch_downsample_wave), so y4[0..256) is zero as well
    const int z4 = zero_hist ? (MAXP >> 3) * 2 : 0;
    for (int j = lane(); j < z4; j += LANES) y4[j] = 0;
    for (int j = z4 + lane(); j < (lag >> 2); j += LANES) { i32 v = y[2 * j]; y4[j] = (i16)v; my = imax(my, v); ny = imin(ny, v); }
    i32 xmax = imax(wave_max(mx), -wave_min(mn));
    i32 ymax = imax(wave_max(my), -wave_min(ny));
    int shift = celt_ilog2(imax(1, imax(xmax, ymax))) - 11;
    wave_sync();
    if (shift > 0) {
        for (int j = lane(); j < (len >> 2); j += LANES) x4[j] = (i16)(x4[j] >> shift);
        for (int j = z4 + lane(); j < (lag >> 2); j += LANES) y4[j] = (i16)(y4[j] >> shift);
        shift *= 2;
    } else {
        shift = 0;
    }
    wave_sync();
    CA_STAMP_F(F, 26);
    // coarse search, 4x decimation: celt_pitch_xcorr(x4, y4, xcorr, 240, 244). Each lane owns four consecutive lags
    // and slides an 8-sample window of y4 along x4 (the shape of xcorr_kernel, pitch.h:61-129): per four samples one
    // 8-byte read of x4 (a broadcast) and one of y4 instead of eight 2-byte reads, and a 60-trip loop instead of 4 x 240.
    // The sums wrap (MAC16_16), so the order of accumulation is free.
    static_assert(((MAXP - 3 * MINP) >> 2) % 4 == 0 && (FRAME >> 2) % 4 == 0, "244 lags = 61 groups of 4; 240 samples = 60 groups of 4");
    i32 mc = 1;
    for (int i0 = 4 * lane(); i0 < (max_pitch >> 2); i0 += 4 * LANES) {
        const int2 *xv = reinterpret_cast<const int2 *>(x4), *yv = reinterpret_cast<const int2 *>(y4 + i0);
        i32 s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        int2 w0 = yv[0];
#if !defined(CA_HOST_EMU)
#pragma unroll 4
#endif
        for (int c = 0; c < (len >> 4); c++) {
            const int2 w1 = yv[c + 1], xx = xv[c];
            const i32 x0 = (i16)xx.x, x1 = xx.x >> 16, x2 = (i16)xx.y, x3 = xx.y >> 16;
            const i32 y0 = (i16)w0.x, y1 = w0.x >> 16, y2 = (i16)w0.y, y3 = w0.y >> 16;
            const i32 y4_ = (i16)w1.x, y5 = w1.x >> 16, y6 = (i16)w1.y;
            s0 = add32(s0, add32(add32(__mul24(x0, y0), __mul24(x1, y1)), add32(__mul24(x2, y2), __mul24(x3, y3))));
            s1 = add32(s1, add32(add32(__mul24(x0, y1), __mul24(x1, y2)), add32(__mul24(x2, y3), __mul24(x3, y4_))));
            s2 = add32(s2, add32(add32(__mul24(x0, y2), __mul24(x1, y3)), add32(__mul24(x2, y4_), __mul24(x3, y5))));
            s3 = add32(s3, add32(add32(__mul24(x0, y3), __mul24(x1, y4_)), add32(__mul24(x2, y5), __mul24(x3, y6))));
            w0 = w1;
        }
        xcorr[i0] = s0; xcorr[i0 + 1] = s1; xcorr[i0 + 2] = s2; xcorr[i0 + 3] = s3;
        mc = imax(imax(mc, imax(s0, s1)), imax(s2, s3));
    }
    i32 maxcorr = wave_max(mc);
    wave_sync();
    CA_STAMP_F(F, 27);
    int best_pitch[2];
    find_best_pitch_wave(xcorr, y4, len >> 2, max_pitch >> 2, best_pitch, 0, maxcorr);
    wave_sync();
    CA_STAMP_F(F, 28);
    // finer search, 2x decimation: only lags within +-2 of the two candidates are evaluated
    for (int i = lane(); i < (max_pitch >> 1); i += LANES) xcorr[i] = 0;
    wave_sync();
    maxcorr = 1;
    for (int cand = 0; cand < 2; cand++) {
        const int bpc = cand ? best_pitch[1] : best_pitch[0];        // (a run-time index would put best_pitch[] in scratch memory)
        for (int i = imax(0, 2 * bpc - 2); i <= imin((max_pitch >> 1) - 1, 2 * bpc + 2); i++) {
            if (cand == 1) { int d0 = i - 2 * best_pitch[0]; if ((d0 < 0 ? -d0 : d0) <= 2) continue; }   // done already
            i32 p = 0;
            for (int j = lane(); j < (len >> 1); j += LANES) p = add32(p, mul16_16(x_lp[j], y[i + j]) >> shift);
            i32 sum = wave_add(p);
            st0(&xcorr[i], imax(-1, sum));
            maxcorr = imax(maxcorr, sum);
        }
    }
    wave_sync();
    CA_STAMP_F(F, 29);
    find_best_pitch_wave(xcorr, y, len >> 1, max_pitch >> 1, best_pitch, shift + 1, maxcorr);
    int offset = 0;
    if (best_pitch[0] > 0 && best_pitch[0] < (max_pitch >> 1) - 1) {
        i32 a = xcorr[best_pitch[0] - 1], b = xcorr[best_pitch[0]], c = xcorr[best_pitch[0] + 1];
        if (sub32(c, a) > mul16_32_q15(22938, sub32(b, a))) offset = 1;            // QCONST16(.7f,15)
        else if (sub32(a, c) > mul16_32_q15(22938, sub32(b, c))) offset = -1;
    }
    wave_sync();
    return 2 * best_pitch[0] - offset;
}

// inner products over the half-rate pitch buffer: sum_{i<N} x[i]*y1[i], x[i]*y2[i]  (pitch.h:132-160)
CA_DEV void dual_inner_prod_wave(const i16 *x, const i16 *y1, const i16 *y2, int N, i32 &xy1, i32 &xy2)
{
    i32 a = 0, b = 0;
    for (int i = lane(); i < N; i += LANES) { a = mac16_16(a, x[i], y1[i]); b = mac16_16(b, x[i], y2[i]); }
    xy1 = wave_add(a);
    xy2 = wave_add(b);
}

CA_DEV i32 pitch_gain_from(i32 xy, i32 xx, i32 yy, bool halve)                      // pitch.c:406-421,:455-460
{
    i32 m = mul32_32_q31(xx, yy);
    i32 x2y2 = add32(1, halve ? (m >> 1) : m);
    int sh = celt_ilog2(x2y2) >> 1;
    i32 t = vshr32(x2y2, 2 * (sh - 7));
    return vshr32(mul16_32_q15(celt_rsqrt_norm(t), xy), sh + 1);
}

// remove_doubling(x = buf, maxperiod 1024, minperiod 15, N 960, &T0, prev_period, prev_gain)  (pitch.c:372-503)
template <class L>
CA_DEVFN i32 remove_doubling_wave(L &F, int *T0_, int prev_period, i32 prev_gain)
{
    const int minperiod0 = MINP, maxperiod = MAXP / 2, minperiod = MINP / 2, N = FRAME / 2;
    const i16 *x = F.s.pitch.buf + maxperiod;
    i32 *yy_lookup = F.s.pitch.xcorr;                                               // 513 entries (xcorr is dead)
    *T0_ /= 2;
    prev_period /= 2;
    if (*T0_ >= maxperiod) *T0_ = maxperiod - 1;
    int T, T0;
    T = T0 = *T0_;
    i32 xx, xy;
    dual_inner_prod_wave(x, x, x - T0, N, xx, xy);
    {   // yy_lookup[i] = max(0, xx + sum_{j=1..i} (x[-j]^2 - x[N-j]^2)): a prefix sum (wrap-around adds commute),
        // scanned LANES entries at a time
        st0(&yy_lookup[0], xx);
        i32 run = xx;
        for (int base = 1; base <= maxperiod; base += LANES) {
            const int i = base + lane();
            i32 d = 0;
            if (i <= maxperiod) d = sub32(mul16_16(x[-i], x[-i]), mul16_16(x[N - i], x[N - i]));
            const i32 inc = wave_scan_add(d);
            if (i <= maxperiod) yy_lookup[i] = imax(0, add32(run, inc));
            run = add32(run, wave_last(inc));
        }
    }
    wave_sync();
    i32 yy = yy_lookup[T0];
    i32 best_xy = xy, best_yy = yy;
    i32 g, g0;
    g = g0 = pitch_gain_from(xy, xx, yy, true);
    for (int k = 2; k <= 15; k++) {
        int T1 = (2 * T0 + k) / (2 * k), T1b;
        if (T1 < minperiod) break;
        if (k == 2) T1b = (T1 + T0 > maxperiod) ? T0 : T0 + T1;
        else T1b = (2 * CLT_second_check[k] * T0 + k) / (2 * k);
        i32 xy2;
        dual_inner_prod_wave(x, x - T1, x - T1b, N, xy, xy2);
        xy = add32(xy, xy2);
        yy = add32(yy_lookup[T1], yy_lookup[T1b]);
        i32 g1 = pitch_gain_from(xy, xx, yy, false);
        int dT = T1 - prev_period;
        if (dT < 0) dT = -dT;
        i32 cont;
        if (dT <= 1) cont = prev_gain;
        else if (dT <= 2 && 5 * k * k < T0) cont = (i16)(prev_gain >> 1);
        else cont = 0;
        i32 thresh = imax(9830, sub32(mul16_32_q15(22938, g0), cont));              // .3, .7
        if (T1 < 3 * minperiod) thresh = imax(13107, sub32(mul16_32_q15(27853, g0), cont));   // .4, .85
        else if (T1 < 2 * minperiod) thresh = imax(16384, sub32(mul16_32_q15(29491, g0), cont));
        if (g1 > thresh) { best_xy = xy; best_yy = yy; T = T1; g = g1; }
    }
    best_xy = imax(0, best_xy);
    i32 pg;
    if (best_yy <= best_xy) pg = 32767;
    else pg = (i16)(frac_div32(best_xy, add32(best_yy, 1)) >> 16);
    i32 xc[3];
    for (int k = 0; k < 3; k++) {
        i32 p = 0;
        const i16 *yk = x - (T + k - 1);
        for (int i = lane(); i < N; i += LANES) p = mac16_16(p, x[i], yk[i]);
        xc[k] = wave_add(p);
    }
    int offset = 0;
    if (sub32(xc[2], xc[0]) > mul16_32_q15(22938, sub32(xc[1], xc[0]))) offset = 1;
    else if (sub32(xc[0], xc[2]) > mul16_32_q15(22938, sub32(xc[1], xc[2]))) offset = -1;
    if (pg > g) pg = (i16)g;
    *T0_ = 2 * T + offset;
    if (*T0_ < minperiod0) *T0_ = minperiod0;
    wave_sync();
    return pg;
}

// comb_filter(y = in[c]+OVL, x = pre[c]+1024, T0, T1, N = 960, g0, g1, tapset0, tapset1, window, 120)
// (celt.c:183-237; the x86 build uses the plain comb_filter_const_c, celt.c:156-181). Pure FIR on the
// unfiltered signal, so every output sample is independent.
template <class L>
CA_DEVFN void comb_filter_wave(L &F, const FrameCtx &fc, int c, int T0, int T1, i32 g0, i32 g1,
                               int tapset0, int tapset1)
{
    if (g0 == 0 && g1 == 0) {
        for (int i = lane(); i < FRAME; i += LANES) tsig(F, c)[OVL + i] = F.xf[c][i];
        return;
    }
    const i16 *G = CLT_comb_gains;
    i32 g00 = (i16)mul16_16_p15(g0, G[tapset0 * 3 + 0]), g01 = (i16)mul16_16_p15(g0, G[tapset0 * 3 + 1]),
        g02 = (i16)mul16_16_p15(g0, G[tapset0 * 3 + 2]);
    i32 g10 = (i16)mul16_16_p15(g1, G[tapset1 * 3 + 0]), g11 = (i16)mul16_16_p15(g1, G[tapset1 * 3 + 1]),
        g12 = (i16)mul16_16_p15(g1, G[tapset1 * 3 + 2]);
    int overlap = OVL;
    if (g0 == g1 && T0 == T1 && tapset0 == tapset1) overlap = 0;
#define CA_PX(k) pre_at(F, fc, c, MAXP + (k))
    for (int i = lane(); i < FRAME; i += LANES) {
        i32 xi = F.xf[c][i];
        i32 y;
        if (i < overlap) {
            i32 w = CLT_window120[i];
            i32 f = (i16)mul16_16_q15(w, w);
            i32 nf = (i16)(32767 - f);
            y = add32(xi, mul16_32_q15((i16)mul16_16_q15(nf, g00), CA_PX(i - T0)));
            y = add32(y, mul16_32_q15((i16)mul16_16_q15(nf, g01), add32(CA_PX(i - T0 + 1), CA_PX(i - T0 - 1))));
            y = add32(y, mul16_32_q15((i16)mul16_16_q15(nf, g02), add32(CA_PX(i - T0 + 2), CA_PX(i - T0 - 2))));
            y = add32(y, mul16_32_q15((i16)mul16_16_q15(f, g10), CA_PX(i - T1)));
            y = add32(y, mul16_32_q15((i16)mul16_16_q15(f, g11), add32(CA_PX(i - T1 + 1), CA_PX(i - T1 - 1))));
            y = add32(y, mul16_32_q15((i16)mul16_16_q15(f, g12), add32(CA_PX(i - T1 + 2), CA_PX(i - T1 - 2))));
        } else if (g1 == 0) {
            y = xi;
        } else {
            y = add32(xi, mul16_32_q15(g10, CA_PX(i - T1)));
            y = add32(y, mul16_32_q15(g11, add32(CA_PX(i - T1 + 1), CA_PX(i - T1 - 1))));
            y = add32(y, mul16_32_q15(g12, add32(CA_PX(i - T1 + 2), CA_PX(i - T1 - 2))));
        }
        tsig(F, c)[OVL + i] = y;
    }
#undef CA_PX
}

struct PrefilterOut { int pf_on, pitch_index, qg; i32 gain1; };

// run_prefilter (celt_encoder.c:1067-1193). `in_mem` = previous frame's last 120 filtered samples (or zero).
template <class L>
CA_DEVFN PrefilterOut run_prefilter_wave(L &F, FrameCtx &fc, const i32 *in_mem, int prefilter_tapset,
                                         int enabled, int nbAvailableBytes, int loss_rate)
{
    const int C = fc.C;
    PrefilterOut o;
    // pre[c] = [history | new]: keep the unfiltered new samples in xf (the split pipeline put them there already)
    if (!L::IN_IS_GLOBAL) {
        for (int c = 0; c < C; c++)
            for (int i = lane(); i < FRAME; i += LANES) F.xf[c][i] = tsig(F, c)[OVL + i];
        wave_sync();
    }
    int pitch_index;
    i32 gain1;
    if (enabled) {
        CA_STAMP_F(F, 3);
        pitch_downsample_wave(F, fc);
        CA_STAMP_F(F, 23);
        pitch_index = pitch_search_wave(F, fc.hist == nullptr);
        CA_STAMP_F(F, 24);
        pitch_index = MAXP - pitch_index;
        gain1 = remove_doubling_wave(F, &pitch_index, fc.prefilter_period, fc.prefilter_gain);
        CA_STAMP_F(F, 25);
        if (pitch_index > MAXP - 2) pitch_index = MAXP - 2;
        gain1 = (i16)mul16_16_q15(22938, gain1);                                    // QCONST16(.7f,15)
        if (loss_rate > 2) gain1 = (i16)(gain1 >> 1);
        if (loss_rate > 4) gain1 = (i16)(gain1 >> 1);
        if (loss_rate > 8) gain1 = 0;
    } else {
        gain1 = 0;
        pitch_index = MINP;
    }
    i32 pf_threshold = 6554;                                                         // QCONST16(.2f,15)
    {
        int d = pitch_index - fc.prefilter_period;
        if (d < 0) d = -d;
        if (d * 10 > pitch_index) pf_threshold += 6554;
    }
    if (nbAvailableBytes < 25) pf_threshold += 3277;
    if (nbAvailableBytes < 35) pf_threshold += 3277;
    if (fc.prefilter_gain > 13107) pf_threshold -= 3277;
    if (fc.prefilter_gain > 18022) pf_threshold -= 3277;
    pf_threshold = imax(pf_threshold, 6554);
    if (gain1 < pf_threshold) {
        gain1 = 0;
        o.pf_on = 0;
        o.qg = 0;
    } else {
        i32 d = gain1 - fc.prefilter_gain;
        if ((d < 0 ? -d : d) < 3277) gain1 = fc.prefilter_gain;
        int qg = ((gain1 + 1536) >> 10) / 3 - 1;
        qg = imax(0, imin(7, qg));
        gain1 = 3072 * (qg + 1);                                                     // QCONST16(0.09375f,15)*(qg+1)
        o.pf_on = 1;
        o.qg = qg;
    }
    fc.prefilter_period = imax(fc.prefilter_period, MINP);
    for (int c = 0; c < C; c++) {
        for (int i = lane(); i < OVL; i += LANES) tsig(F, c)[i] = in_mem ? in_mem[c * OVL + i] : 0;
        // shortMdctSize - overlap == 0, so only the cross-faded filter runs (celt_encoder.c:1168-1176)
        comb_filter_wave(F, fc, c, fc.prefilter_period, pitch_index, (i16)neg32(fc.prefilter_gain),
                         (i16)neg32(gain1), fc.prefilter_tapset, prefilter_tapset);
    }
    wave_sync();
    o.gain1 = gain1;
    o.pitch_index = pitch_index;
    return o;
}

// ---- transient_analysis (celt_encoder.c:227-377), len = 1080 ---------------------------------------
struct TransientOut { int is_transient, tf_chan; i32 tf_estimate; };

// channel with the largest masking metric decides (celt_encoder.c:352-375)
CA_DEV TransientOut transient_combine(const i32 *unmask, int C)
{
    TransientOut o;
    o.tf_chan = 0;
    i32 mask_metric = 0;
    for (int c = 0; c < C; c++)
        if (unmask[c] > mask_metric) { o.tf_chan = c; mask_metric = unmask[c]; }
    o.is_transient = mask_metric > 200;
    i32 tf_max = imax(0, (i16)(celt_sqrt(27 * mask_metric) - 42));
    o.tf_estimate = (i16)celt_sqrt(imax(0, sub32(shl32(mul16_16(113, imin(163, tf_max)), 14), 37312528)));   // .0069 Q14, .139 Q28
    return o;
}

template <class L>
CA_DEVFN TransientOut transient_analysis_wave(L &F, const FrameCtx &fc)
{
    const int C = fc.C, len = FRAME + OVL, len2 = len / 2;
    // serial IIR/followers: lane c owns channel c
    for (int c = lane(); c < C; c += LANES) {
        i16 *tmp = F.s.trans[c];
        i32 mem0 = 0, mem1 = 0;
        for (int i = 0; i < len; i++) {
            i32 x = tsig(F, c)[i] >> 12;
            i32 y = add32(mem0, x);
            mem0 = sub32(add32(mem1, y), shl32(x, 1));
            mem1 = sub32(x, y >> 1);
            tmp[i] = (i16)(y >> 2);
        }
        for (int i = 0; i < 12; i++) tmp[i] = 0;
        i32 mx = 0, mn = 0;
        for (int i = 0; i < len; i++) { mx = imax(mx, tmp[i]); mn = imin(mn, tmp[i]); }
        int shift = 14 - celt_ilog2(1 + imax(mx, -mn));
        if (shift != 0)
            for (int i = 0; i < len; i++) tmp[i] = (i16)shl16(tmp[i], shift);
        i32 mean = 0;
        mem0 = 0;
        for (int i = 0; i < len2; i++) {
            i32 x2 = (i16)pshr32(add32(mul16_16(tmp[2 * i], tmp[2 * i]), mul16_16(tmp[2 * i + 1], tmp[2 * i + 1])), 16);
            mean = add32(mean, x2);
            tmp[i] = (i16)(mem0 + pshr32(x2 - mem0, 4));
            mem0 = tmp[i];
        }
        mem0 = 0;
        i32 maxE = 0;
        for (int i = len2 - 1; i >= 0; i--) {
            tmp[i] = (i16)(mem0 + pshr32(tmp[i] - mem0, 3));
            mem0 = tmp[i];
            maxE = imax(maxE, mem0);
        }
        mean = mul16_16(celt_sqrt(mean), celt_sqrt(mul16_16(maxE, len2 >> 1)));
        i32 norm = shl32(len2, 6 + 14) / add32(1, mean >> 1);
        F.scal[4 + c] = norm;
    }
    wave_sync();
    i32 um[2] = {0, 0};
    for (int c = 0; c < C; c++) {
        const i16 *tmp = F.s.trans[c];
        i32 norm = F.scal[4 + c];
        i32 p = 0;
        for (int k = lane(); 12 + 4 * k < len2 - 5; k += LANES) {
            int i = 12 + 4 * k;
            i32 id = imax(0, imin(127, mul16_32_q15((i16)(tmp[i] + 1), norm)));
            p += CLT_inv_table[id];
        }
        i32 unmask = wave_add(p);
        um[c] = 64 * unmask * 4 / (6 * (len2 - 17));
    }
    wave_sync();
    return transient_combine(um, C);
}

// ---- MDCTs of one frame (compute_mdcts, celt_encoder.c:418-461): in -> xf (as freq) ------------------
// where channel c's coefficients live: its own row, or THE row of a one-channel working set
template <class L> CA_DEV i32 *xf_row(L &F, int c) { return F.xf[L::XF_CHANNELS == 1 ? 0 : c]; }

template <class L>
CA_DEVFN void compute_mdct_channel(L &F, int c, int shortBlocks)
{
    if (shortBlocks) {
        const MdctTab T = mdct_global_tab<3>();
        mdct_forward_wave<3, 8>(tsig(F, c), F.s.f2, xf_row(F, c), 1, T, lane());
    } else {
        const MdctTab T = mdct_global_tab<0>();
        mdct_forward_wave<0, 1>(tsig(F, c), F.s.f2, xf_row(F, c), 1, T, lane());
    }
}

template <class L>
CA_DEVFN void compute_mdcts_wave(L &F, const FrameCtx &fc, int shortBlocks)
{
    for (int c = 0; c < fc.C; c++) compute_mdct_channel(F, c, shortBlocks);
}

// ---- compute_band_energies + amp2Log2 (bands.c:97-142, quant_bands.c:551-575) -----------------------
// Channel c alone, over all 64 lanes: a band's two reductions (largest magnitude, then the sum of squares at the shift that
// magnitude fixes) are split into chunks of eight bins -- 100 chunks per channel, one lane each -- and combined per band by
// the band's own lane: 8 + 22 dependent steps per reduction instead of the 176 of the widest band (maxima compose exactly,
// the sums wrap, so the order of the adds is free). The partial results sit in the FFT scratch, idle between transforms.
template <class L>
CA_DEVFN void band_energies_channel(L &F, int c, i16 *bandLogE)
{
    const i32 *X = xf_row(F, c);
    i32 *red = reinterpret_cast<i32 *>(F.s.f2);                 // [0,100) chunk max, [100,200) chunk min, [200,300) chunk sums, [300,321) band shift
    constexpr int NCH = 100;                                    // CLT_eband5ms[NB] chunks of eight bins (LM 3)
    // (a lane reads its chunk's eight consecutive words -- 32-byte stride across lanes, some bank conflicts. The conflict-free
    // alternative, one bin per lane and a cross-lane reduction per octet, was measured SLOWER, 1.79 -> 2.07 ms on the whole kernel
    // with DPP reductions, 2.16 with ds_bpermute ones: this kernel is bound by vector-ALU issue, not by the LDS.)
    for (int ch = lane(); ch < NCH; ch += LANES) {
        i32 mx = 0, mn = 0;
#pragma unroll
        for (int u = 0; u < 8; u++) { const i32 v = X[8 * ch + u]; mx = imax(mx, v); mn = imin(mn, v); }
        red[ch] = mx;
        red[NCH + ch] = mn;
    }
    wave_sync();
    for (int b = lane(); b < NB; b += LANES) {
        i32 mx = 0, mn = 0;
        for (int ch = CLT_eband5ms[b]; ch < CLT_eband5ms[b + 1]; ch++) { mx = imax(mx, red[ch]); mn = imin(mn, red[NCH + ch]); }
        const i32 maxval = imax(mx, neg32(mn));
        // shift of the band, or "no energy" (bands.c:108-126)
        red[3 * NCH + b] = maxval > 0 ? celt_ilog2(maxval) - 14 + (((CLT_logN400[b] >> 3) + LM3 + 1) >> 1) : 0x7fff;
    }
    wave_sync();
    for (int ch = lane(); ch < NCH; ch += LANES) {
        const int shift = red[3 * NCH + CLT_bin2band[ch]];
        i32 sum = 0;
        if (shift != 0x7fff) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const i32 x = X[8 * ch + u];
                const i32 v = shift > 0 ? (i16)(x >> shift) : (i16)shl32(x, -shift);
                sum = mac16_16(sum, v, v);
            }
        }
        red[2 * NCH + ch] = sum;
    }
    wave_sync();
    for (int b = lane(); b < NB; b += LANES) {
        const int shift = red[3 * NCH + b];
        i32 E = 1;
        if (shift != 0x7fff) {
            i32 sum = 0;
            for (int ch = CLT_eband5ms[b]; ch < CLT_eband5ms[b + 1]; ch++) sum = add32(sum, red[2 * NCH + ch]);
            E = add32(1, vshr32(celt_sqrt(sum), -shift));
        }
        F.bandE[c * NB + b] = E;
        bandLogE[c * NB + b] = (i16)(celt_log2(shl32(E, 2)) - shl16(CLT_eMeans[b], 6));
    }
    wave_sync();
}

template <class L>
CA_DEVFN void band_energies_wave(L &F, const FrameCtx &fc, i16 *bandLogE)
{
    for (int c = 0; c < fc.C; c++) band_energies_channel(F, c, bandLogE);
}

// ---- normalise_bands (bands.c:146-168): xf (freq) -> X (int16, aliases `in`) ------------------------
template <class L>
CA_DEVFN void normalise_bands_channel(L &F, int c)
{
    i16 *X = frame_X(F);
    const i32 *xf = xf_row(F, c);
    for (int k = c * NB + lane(); k < (c + 1) * NB; k += LANES) {
        i32 bE = F.bandE[k];
        int shift = celt_zlog2(bE) - 13;
        i32 E = (i16)vshr32(bE, shift);                                              // opus_val16 E
        F.normg[k] = (i16)celt_rcp(shl32(E, 3));
        F.normshift[k] = (i8)shift;
    }
    wave_sync();
    for (int j = lane(); j < (CLT_eband5ms[NB] << LM3); j += LANES) {
        int b = CLT_bin2band[j >> 3] + c * NB;
        X[c * FRAME + j] = (i16)mul16_16_q15(vshr32(xf[j], F.normshift[b] - 1), F.normg[b]);
    }
    wave_sync();
}

template <class L>
CA_DEVFN void normalise_bands_wave(L &F, const FrameCtx &fc)
{
    for (int c = 0; c < fc.C; c++) normalise_bands_channel(F, c);
}

}  // namespace ca

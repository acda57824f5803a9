// tests/emu/celt_emu.cpp -- TEST INFRASTRUCTURE: host build of the frame-kernel sources with
// CA_HOST_EMU (one "lane"), so the kernel logic can be single-stepped and compared against the
// reference on a CPU. Not a product path: concentus_amd/ never loads this library.
#define CA_HOST_EMU 1
#include <stdlib.h>
#include <string.h>
#include <map>
#include <string>
#include <vector>
static std::map<std::string, std::vector<unsigned char>> g_taps;
extern "C" void emu_tap(const char *name, const void *p, int bytes)
{
    g_taps[name].assign((const unsigned char *)p, (const unsigned char *)p + bytes);
}
extern "C" int emu_tap_get(const char *name, void *out, int cap)
{
    auto it = g_taps.find(name);
    if (it == g_taps.end()) return -1;
    int n = (int)it->second.size();
    memcpy(out, it->second.data(), n < cap ? n : cap);
    return n;
}
#include "../../concentus_amd/csrc/celt_enc.h"
#include "../../concentus_amd/csrc/celt_stage_lane.h"
#include "../../concentus_amd/csrc/celt_dec.h"

using namespace ca;

// ---- workload counters (CA_COUNT) ----
#include <map>
#include <string>
static std::map<std::string, std::pair<long, long>> g_counts;       // name -> (events, sum)
extern "C" void emu_count(const char *name, long n) { auto &c = g_counts[name]; c.first++; c.second += n; }
extern "C" void emu_counts_reset(void) { g_counts.clear(); }
extern "C" int emu_counts_dump(char *buf, int cap)
{
    std::string o;
    for (auto &kv : g_counts) {
        char line[160];
        snprintf(line, sizeof line, "%s %ld %ld\n", kv.first.c_str(), kv.second.first, kv.second.second);
        o += line;
    }
    if ((int)o.size() + 1 > cap) return -1;
    memcpy(buf, o.c_str(), o.size() + 1);
    return (int)o.size();
}

extern "C" int emu_celt_encode_frames(const opusgpu_celt_config *cfg, opusgpu_celt_state *states /* or NULL */,
                                      const int16_t *pcm, int nframes, int frames_per_stream,
                                      unsigned char *out, int out_stride, int *out_len, uint32_t *out_rng)
{
    // frames are laid out stream-major: frame f of stream s at index s*frames_per_stream + f.
    // Runs the same two phases as the GPU kernels (front -> FrameMid -> back).
    FrontLds *F1 = (FrontLds *)aligned_alloc(64, sizeof(FrontLds) + 64);
    BackLds *F2 = (BackLds *)aligned_alloc(64, sizeof(BackLds) + 64);
    FrameMid *mid = (FrameMid *)aligned_alloc(64, sizeof(FrameMid) + 64);
    const int C = cfg->channels;
    for (int n = 0; n < nframes; n++) {
        const int poison = getenv("EMU_POISON") ? atoi(getenv("EMU_POISON")) : 0xAB;
        memset(F1, poison, sizeof(FrontLds));     // poison: catch reads of never-written LDS
        memset(F2, poison, sizeof(BackLds));
        memset(mid, 0xCD, sizeof(FrameMid));
        opusgpu_celt_state *st = states ? &states[n / frames_per_stream] : NULL;
        celt_encode_front(*F1, *cfg, st, st, pcm + (size_t)n * 960 * C, mid);
        FrameResult r = celt_encode_back(*F2, *cfg, mid, st, out + (size_t)n * out_stride);
        out_len[n] = r.bytes;
        out_rng[n] = r.final_range;
    }
    free(F1);
    free(F2);
    free(mid);
    return 0;
}

// The split pipeline of the GPU library on the CPU: dc_reject stage -> front phase 1 -> transient stage ->
// front phase 2 -> back phase, with the same hand-off buffers.
extern "C" int emu_celt_encode_frames_split(const opusgpu_celt_config *cfg, opusgpu_celt_state *states /* or NULL */,
                                            const int16_t *pcm, int nframes, int frames_per_stream,
                                            unsigned char *out, int out_stride, int *out_len, uint32_t *out_rng)
{
    FrontLds *F1 = (FrontLds *)aligned_alloc(64, sizeof(FrontLds) + 64);
    BackLds *F2 = (BackLds *)aligned_alloc(64, sizeof(BackLds) + 64);
    FrameMid *mid = (FrameMid *)aligned_alloc(64, sizeof(FrameMid) + 64);
    int32_t *in_ws = (int32_t *)aligned_alloc(64, 2 * 1080 * 4);
    const int C = cfg->channels;
    for (int n = 0; n < nframes; n++) {
        memset(F1, 0xAB, sizeof(FrontLds));
        memset(F2, 0xAB, sizeof(BackLds));
        memset(mid, 0xCD, sizeof(FrameMid));
        memset(in_ws, 0xEF, 2 * 1080 * 4);
        opusgpu_celt_state *st = states ? &states[n / frames_per_stream] : NULL;
        const int16_t *p = pcm + (size_t)n * 960 * C;
        for (int c = 0; c < C; c++) {
            i32 hp[2] = {st ? st->hp_mem[2 * c] : 0, st ? st->hp_mem[2 * c + 1] : 0};
            stage_dc_reject_channel(p, c, hp, mid->X + c * 960);
            mid->hp_mem[2 * c] = hp[0];
            mid->hp_mem[2 * c + 1] = hp[1];
        }
        Front1Lds *Fa = (Front1Lds *)F1;                 // the split kernels' own (smaller) working sets
        Fa->in_g = in_ws;
        celt_encode_front_phase<1>(*Fa, *cfg, st, st, p, mid, nullptr, in_ws);
        for (int c = 0; c < C; c++) mid->trans_unmask[c] = stage_transient_channel(in_ws + c * 1080, mid->X + c * 960);
        memset(F1, 0xAB, sizeof(FrontLds));
        Front2Lds *Fb = (Front2Lds *)F1;
        Fb->in_g = in_ws;
        Fb->x_g = mid->X;
        celt_encode_front_phase<2>(*Fb, *cfg, nullptr, nullptr, nullptr, mid, nullptr, in_ws);
        FrameResult r = celt_encode_back(*F2, *cfg, mid, st, out + (size_t)n * out_stride);
        out_len[n] = r.bytes;
        out_rng[n] = r.final_range;
    }
    free(F1); free(F2); free(mid); free(in_ws);
    return 0;
}

// ---- decoder: packets -> PCM, streams of frames_per_stream packets each starting from a fresh decoder ----
static void dec_state_reset(opusgpu_celt_dec_state *st)
{
    memset(st, 0, sizeof(*st));
    for (int i = 0; i < 42; i++) st->oldLogE[i] = st->oldLogE2[i] = -28672;
}

extern "C" int emu_celt_decode_frames(const unsigned char *packets, int stride, const int *len, int nframes,
                                      int frames_per_stream, int16_t *pcm, uint32_t *rng, int *ret)
{
    DecWork *F = (DecWork *)aligned_alloc(64, sizeof(DecWork) + 64);
    SynthLds *L = (SynthLds *)aligned_alloc(64, sizeof(SynthLds) + 64);
    opusgpu_celt_dec_state *st = (opusgpu_celt_dec_state *)aligned_alloc(64, sizeof(opusgpu_celt_dec_state) + 64);
    for (int n = 0; n < nframes; n++) {
        if (n % frames_per_stream == 0) dec_state_reset(st);
        memset(F, 0xAB, sizeof(DecWork));
        memset(L, 0xAB, sizeof(SynthLds));
        DecResult r = celt_decode_frame(*F, *L, st, packets + (size_t)n * stride, len[n], pcm + (size_t)n * 960 * 2);
        ret[n] = r.samples;
        rng[n] = r.final_range;
    }
    free(F);
    free(L);
    free(st);
    return 0;
}
extern "C" int emu_sizeof_dec_state(void) { return (int)sizeof(opusgpu_celt_dec_state); }

extern "C" int emu_sizeof_state(void) { return (int)sizeof(opusgpu_celt_state); }
extern "C" int emu_sizeof_frame_lds(void) { return (int)(sizeof(FrontLds) * 100000 + sizeof(BackLds)); }

// MDCT device functions under emulation (same sources as the MDCT-only kernels)
extern "C" void emu_mdct_forward(const int32_t *in, int32_t *out, int shift)
{
    static int2 f2[480];
    if (shift == 0) { MdctTab T = mdct_global_tab<0>(); mdct_forward_wave<0, 1>(in, f2, out, 1, T, 0); }
    else { MdctTab T = mdct_global_tab<3>(); mdct_forward_wave<3, 8>(in, f2, out, 1, T, 0); }
}
extern "C" void emu_mdct_backward(const int32_t *coef, int32_t *out, int shift)
{
    static int2 f2[480];
    if (shift == 0) { MdctTab T = mdct_global_tab<0>(); mdct_backward_wave<0, 1>(coef, 1, f2, out, T, 0); }
    else { MdctTab T = mdct_global_tab<3>(); mdct_backward_wave<3, 8>(coef, 1, f2, out, T, 0); }
}

// ---- silk_find_LPC_FIX, host build of concentus_amd/csrc/silk_lpc_dev.h (CPU tier of tests/test_silk_lpc_cpu.py) ----
#include "../../concentus_amd/csrc/silk_lpc_dev.h"
#include "../../include/opusgpu_silk.h"
extern "C" void emu_silk_find_lpc(const opusgpu_find_lpc_in *in, opusgpu_find_lpc_out *out, long n)
{
    for (long r = 0; r < n; r++) {
        int16_t nlsf[16] = {0};
        out[r].NLSFInterpCoef_Q2 = ca::silk_find_LPC_dev(in[r].x, in[r].minInvGain_Q30, in[r].subfr_length, in[r].nb_subfr,
                                                         in[r].predictLPCOrder, in[r].useInterpolatedNLSFs, in[r].first_frame_after_reset,
                                                         in[r].prev_NLSFq_Q15, nlsf);
        for (int k = 0; k < 16; k++) out[r].NLSF_Q15[k] = nlsf[k];
        out[r].status = 0;
    }
}

// ---- silk_process_NLSFs / silk_residual_energy_FIX, host build of concentus_amd/csrc/silk_nlsf_dev.h (tests/test_silk_nlsf_cpu.py) ----
#include "../../concentus_amd/csrc/silk_nlsf_dev.h"
extern "C" void emu_silk_process_nlsfs(const opusgpu_process_nlsf_in *in, opusgpu_process_nlsf_out *out, long n)
{
    for (long r = 0; r < n; r++) {
        int16_t nlsf[16], pc[2][16];
        int8_t idx[17];
        memset(pc, 0, sizeof(pc));
        memset(idx, 0, sizeof(idx));
        for (int k = 0; k < 16; k++) nlsf[k] = in[r].NLSF_Q15[k];
        ca::silk_process_NLSFs_dev(pc, idx, nlsf, in[r].prev_NLSFq_Q15, in[r].speech_activity_Q8, in[r].nb_subfr, in[r].predictLPCOrder,
                                   in[r].useInterpolatedNLSFs, in[r].NLSFInterpCoef_Q2, in[r].NLSF_MSVQ_Survivors, in[r].signalType);
        memset(&out[r], 0, sizeof(out[r]));
        for (int k = 0; k < in[r].predictLPCOrder; k++) {
            out[r].PredCoef_Q12[0][k] = pc[0][k]; out[r].PredCoef_Q12[1][k] = pc[1][k]; out[r].NLSF_Q15[k] = nlsf[k];
        }
        for (int k = 0; k <= in[r].predictLPCOrder; k++) out[r].NLSFIndices[k] = idx[k];
    }
}
extern "C" void emu_silk_residual_energy(const opusgpu_res_nrg_in *in, opusgpu_res_nrg_out *out, long n)
{
    for (long r = 0; r < n; r++) {
        int32_t nrgs[4] = {0, 0, 0, 0}, nrgsQ[4] = {0, 0, 0, 0};
        ca::silk_residual_energy_dev(nrgs, nrgsQ, in[r].x, in[r].a_Q12, in[r].gains, in[r].subfr_length, in[r].nb_subfr, in[r].LPC_order);
        memset(&out[r], 0, sizeof(out[r]));
        for (int k = 0; k < in[r].nb_subfr; k++) { out[r].nrgs[k] = nrgs[k]; out[r].nrgsQ[k] = nrgsQ[k]; }
    }
}

// ---- silk_find_pred_coefs_FIX, host build of concentus_amd/csrc/silk_pred_dev.h (tests/test_silk_pred_cpu.py) ----
#include "../../concentus_amd/csrc/silk_pred_dev.h"
static void pred_cfg_from_record(const opusgpu_find_pred_coefs_in &in, ca::PredCoefsCfg &c)
{
    for (int k = 0; k < 4; k++) { c.Gains_Q16[k] = in.Gains_Q16[k]; c.pitchL[k] = in.pitchL[k]; }
    for (int k = 0; k < 16; k++) c.prev_NLSFq_Q15[k] = in.prev_NLSFq_Q15[k];
    c.nb_subfr = in.nb_subfr; c.subfr_length = in.subfr_length; c.predictLPCOrder = in.predictLPCOrder; c.ltp_mem_length = in.ltp_mem_length;
    c.signalType = in.signalType; c.condCoding = in.condCoding; c.first_frame_after_reset = in.first_frame_after_reset;
    c.useInterpolatedNLSFs = in.useInterpolatedNLSFs; c.speech_activity_Q8 = in.speech_activity_Q8;
    c.NLSF_MSVQ_Survivors = in.NLSF_MSVQ_Survivors; c.mu_LTP_Q9 = in.mu_LTP_Q9; c.LTPQuantLowComplexity = in.LTPQuantLowComplexity;
    c.sum_log_gain_Q7 = in.sum_log_gain_Q7; c.coding_quality_Q14 = in.coding_quality_Q14; c.PacketLoss_perc = in.PacketLoss_perc;
    c.nFramesPerPacket = in.nFramesPerPacket;
}
static void pred_out_to_record(const ca::PredCoefsOut &o, int order, int nb, opusgpu_find_pred_coefs_out &r)
{
    memset(&r, 0, sizeof(r));
    for (int k = 0; k < order; k++) { r.PredCoef_Q12[0][k] = o.PredCoef_Q12[0][k]; r.PredCoef_Q12[1][k] = o.PredCoef_Q12[1][k]; r.NLSF_Q15[k] = o.NLSF_Q15[k]; }
    for (int k = 0; k < nb * 5; k++) r.LTPCoef_Q14[k] = o.LTPCoef_Q14[k];
    for (int k = 0; k < nb; k++) { r.ResNrg[k] = o.ResNrg[k]; r.ResNrgQ[k] = o.ResNrgQ[k]; r.LTPIndex[k] = o.LTPIndex[k]; }
    r.LTPredCodGain_Q7 = o.LTPredCodGain_Q7; r.LTP_scale_Q14 = o.LTP_scale_Q14; r.sum_log_gain_Q7 = o.sum_log_gain_Q7;
    for (int k = 0; k <= order; k++) r.NLSFIndices[k] = o.NLSFIndices[k];
    r.NLSFInterpCoef_Q2 = (int8_t)o.NLSFInterpCoef_Q2; r.PERIndex = (int8_t)o.PERIndex; r.LTP_scaleIndex = (int8_t)o.LTP_scaleIndex;
}
extern "C" void emu_silk_find_pred_coefs(const opusgpu_find_pred_coefs_in *in, opusgpu_find_pred_coefs_out *out, long n)
{
    for (long r = 0; r < n; r++) {
        ca::PredCoefsCfg c;
        ca::PredCoefsOut o;
        memset(&o, 0, sizeof(o));
        pred_cfg_from_record(in[r], c);
        int16_t pre[OPUSGPU_SILK_BURG_MAX_X];
        memset(pre, 0, sizeof(pre));
        ca::silk_find_pred_coefs_dev(c, (const int16_t *)in[r].res_pitch, (const int16_t *)in[r].x + in[r].ltp_mem_length, (int16_t *)pre, o);
        pred_out_to_record(o, in[r].predictLPCOrder, in[r].nb_subfr, out[r]);
    }
}

// ---- silk_process_gains_FIX, host build of concentus_amd/csrc/silk_gains_dev.h ----
#include "../../concentus_amd/csrc/silk_gains_dev.h"
extern "C" void emu_silk_process_gains(const opusgpu_process_gains_in *in, opusgpu_process_gains_out *out, long n)
{
    for (long r = 0; r < n; r++) {
        const opusgpu_process_gains_in &i = in[r];
        ca::ProcessGainsIO g;
        memset(&g, 0, sizeof(g));
        for (int k = 0; k < 4; k++) { g.Gains_Q16[k] = i.Gains_Q16[k]; g.ResNrg[k] = i.ResNrg[k]; g.ResNrgQ[k] = i.ResNrgQ[k]; }
        g.LastGainIndex = i.LastGainIndex; g.quantOffsetType = i.quantOffsetType;
        ca::silk_process_gains_dev(g, i.signalType, i.nb_subfr, i.subfr_length, i.LTPredCodGain_Q7, i.SNR_dB_Q7, i.condCoding, i.input_tilt_Q15,
                                   i.nStatesDelayedDecision, i.speech_activity_Q8, i.input_quality_Q14, i.coding_quality_Q14);
        memset(&out[r], 0, sizeof(out[r]));
        for (int k = 0; k < i.nb_subfr; k++) { out[r].Gains_Q16[k] = g.Gains_Q16[k]; out[r].GainsUnq_Q16[k] = g.GainsUnq_Q16[k]; out[r].GainsIndices[k] = g.GainsIndices[k]; }
        out[r].Lambda_Q10 = g.Lambda_Q10; out[r].LastGainIndex = g.LastGainIndex; out[r].lastGainIndexPrev = g.lastGainIndexPrev;
        out[r].quantOffsetType = g.quantOffsetType;
    }
}

// ---- silk_noise_shape_analysis_FIX, host build of concentus_amd/csrc/silk_shape_dev.h ----
#include "../../concentus_amd/csrc/silk_shape_dev.h"
static void shape_cfg_from_record(const opusgpu_noise_shape_in &i, ca::ShapeCfg &c, ca::ShapeOut &o)
{
    c.fs_kHz = i.fs_kHz; c.nb_subfr = i.nb_subfr; c.subfr_length = i.subfr_length; c.la_shape = i.la_shape; c.shapeWinLength = i.shapeWinLength;
    c.shapingLPCOrder = i.shapingLPCOrder; c.warping_Q16 = i.warping_Q16; c.SNR_dB_Q7 = i.SNR_dB_Q7; c.useCBR = i.useCBR;
    c.speech_activity_Q8 = i.speech_activity_Q8; c.signalType = i.signalType; c.input_quality_bands_Q15[0] = i.input_quality_bands_Q15[0];
    c.input_quality_bands_Q15[1] = i.input_quality_bands_Q15[1]; c.LTPCorr_Q15 = i.LTPCorr_Q15; c.predGain_Q16 = i.predGain_Q16;
    for (int k = 0; k < 4; k++) c.pitchL[k] = i.pitchL[k];
    memset(&o, 0, sizeof(o));
    o.HarmBoost_smth_Q16 = i.HarmBoost_smth_Q16; o.HarmShapeGain_smth_Q16 = i.HarmShapeGain_smth_Q16; o.Tilt_smth_Q16 = i.Tilt_smth_Q16;
}
static void shape_out_to_record(const ca::ShapeOut &o, int nb, opusgpu_noise_shape_out &r)
{
    memset(&r, 0, sizeof(r));
    for (int k = 0; k < nb; k++) { r.Gains_Q16[k] = o.Gains_Q16[k]; r.GainsPre_Q14[k] = o.GainsPre_Q14[k]; r.LF_shp_Q14[k] = o.LF_shp_Q14[k]; }
    for (int k = 0; k < 64; k++) { r.AR1_Q13[k] = o.AR1_Q13[k]; r.AR2_Q13[k] = o.AR2_Q13[k]; }
    for (int k = 0; k < 4; k++) { r.HarmBoost_Q14[k] = o.HarmBoost_Q14[k]; r.HarmShapeGain_Q14[k] = o.HarmShapeGain_Q14[k]; r.Tilt_Q14[k] = o.Tilt_Q14[k]; }
    r.HarmBoost_smth_Q16 = o.HarmBoost_smth_Q16; r.HarmShapeGain_smth_Q16 = o.HarmShapeGain_smth_Q16; r.Tilt_smth_Q16 = o.Tilt_smth_Q16;
    r.input_quality_Q14 = o.input_quality_Q14; r.coding_quality_Q14 = o.coding_quality_Q14; r.sparseness_Q8 = o.sparseness_Q8;
    r.quantOffsetType = o.quantOffsetType;
}
extern "C" void emu_silk_noise_shape_analysis(const opusgpu_noise_shape_in *in, opusgpu_noise_shape_out *out, long n)
{
    for (long r = 0; r < n; r++) {
        ca::ShapeCfg c;
        ca::ShapeOut o;
        shape_cfg_from_record(in[r], c, o);
        int16_t xw[256], xs[256];
        ca::silk_noise_shape_analysis_dev(c, (const int16_t *)in[r].pitch_res, (const int16_t *)in[r].x + in[r].la_shape, (int16_t *)xw, (int16_t *)xs, o);
        shape_out_to_record(o, in[r].nb_subfr, out[r]);
    }
}

// ---- silk_prefilter_FIX, host build of concentus_amd/csrc/silk_prefilter_dev.h ----
#include "../../concentus_amd/csrc/silk_prefilter_dev.h"
extern "C" void emu_silk_prefilter(const opusgpu_prefilter_in *in, opusgpu_prefilter_state *st, opusgpu_prefilter_out *out, long n)
{
    for (long r = 0; r < n; r++) {
        const opusgpu_prefilter_in &i = in[r];
        ca::PrefilterState P;
        for (int k = 0; k <= 16; k++) P.sAR_shp[k] = st[r].sAR_shp[k];
        P.sLTP_shp_buf_idx = st[r].sLTP_shp_buf_idx; P.sLF_AR_shp_Q12 = st[r].sLF_AR_shp_Q12; P.sLF_MA_shp_Q12 = st[r].sLF_MA_shp_Q12;
        P.sHarmHP_Q2 = st[r].sHarmHP_Q2; P.rand_seed = st[r].rand_seed; P.lagPrev = st[r].lagPrev;
        ca::PrefilterCtrl c;
        for (int k = 0; k < 4; k++) {
            c.pitchL[k] = i.pitchL[k]; c.HarmShapeGain_Q14[k] = i.HarmShapeGain_Q14[k]; c.HarmBoost_Q14[k] = i.HarmBoost_Q14[k];
            c.Tilt_Q14[k] = i.Tilt_Q14[k]; c.GainsPre_Q14[k] = i.GainsPre_Q14[k]; c.LF_shp_Q14[k] = i.LF_shp_Q14[k];
        }
        for (int k = 0; k < 64; k++) c.AR1_Q13[k] = i.AR1_Q13[k];
        c.coding_quality_Q14 = i.coding_quality_Q14; c.nb_subfr = i.nb_subfr; c.subfr_length = i.subfr_length; c.signalType = i.signalType;
        c.warping_Q16 = i.warping_Q16; c.shapingLPCOrder = i.shapingLPCOrder;
        memset(&out[r], 0, sizeof(out[r]));
        ca::silk_prefilter_dev(P, c, (const int16_t *)i.x, (int32_t *)out[r].xw_Q3, (int16_t *)st[r].sLTP_shp);
        for (int k = 0; k <= 16; k++) st[r].sAR_shp[k] = P.sAR_shp[k];
        st[r].sLTP_shp_buf_idx = P.sLTP_shp_buf_idx; st[r].sLF_AR_shp_Q12 = P.sLF_AR_shp_Q12; st[r].sLF_MA_shp_Q12 = P.sLF_MA_shp_Q12;
        st[r].sHarmHP_Q2 = P.sHarmHP_Q2; st[r].lagPrev = P.lagPrev;
    }
}

// ---- silk_find_pitch_lags_FIX, host build of concentus_amd/csrc/silk_pitch_dev.h ----
#include "../../concentus_amd/csrc/silk_pitch_dev.h"
extern "C" void emu_silk_find_pitch_lags(const opusgpu_find_pitch_lags_in *in, opusgpu_find_pitch_lags_out *out, long n)
{
    for (long r = 0; r < n; r++) {
        const opusgpu_find_pitch_lags_in &i = in[r];
        ca::PitchCfg c;
        c.fs_kHz = i.fs_kHz; c.nb_subfr = i.nb_subfr; c.frame_length = i.frame_length; c.ltp_mem_length = i.ltp_mem_length; c.la_pitch = i.la_pitch;
        c.pitch_LPC_win_length = i.pitch_LPC_win_length; c.pitchEstimationLPCOrder = i.pitchEstimationLPCOrder;
        c.pitchEstimationComplexity = i.pitchEstimationComplexity; c.pitchEstimationThreshold_Q16 = i.pitchEstimationThreshold_Q16;
        c.signalType = i.signalType; c.first_frame_after_reset = i.first_frame_after_reset; c.speech_activity_Q8 = i.speech_activity_Q8;
        c.prevSignalType = i.prevSignalType; c.input_tilt_Q15 = i.input_tilt_Q15; c.prevLag = i.prevLag; c.LTPCorr_Q15 = i.LTPCorr_Q15;
        ca::PitchOut o;
        memset(&o, 0, sizeof(o));
        memset(&out[r], 0, sizeof(out[r]));
        int16_t work[640];                 // one array for the window block, its down-shifted copy and the estimator's frame, as in the kernel
        ca::silk_find_pitch_lags_dev(c, (const int16_t *)i.x_buf, (int16_t *)out[r].res, (int16_t *)work, (int16_t *)work, (int16_t *)work, o);
        for (int k = 0; k < i.nb_subfr; k++) out[r].pitchL[k] = o.pitchL[k];
        out[r].lagIndex = o.lagIndex; out[r].contourIndex = o.contourIndex; out[r].LTPCorr_Q15 = o.LTPCorr_Q15; out[r].signalType = o.signalType;
        out[r].predGain_Q16 = o.predGain_Q16;
    }
}

// ---- silk_encode_indices / silk_encode_pulses, host build of concentus_amd/csrc/silk_bits_dev.h ----
#include "../../concentus_amd/csrc/silk_bits_dev.h"
extern "C" void emu_silk_encode_bits(const opusgpu_silk_bits_in *in, opusgpu_ec_state *ecs, opusgpu_silk_bits_out *out, long n)
{
    for (long r = 0; r < n; r++) {
        const opusgpu_silk_bits_in &i = in[r];
        opusgpu_ec_state &st = ecs[r];
        ca::RangeEnc ec;
        ec.buf = st.buf; ec.storage = st.storage; ec.end_offs = st.end_offs; ec.end_window = st.end_window; ec.nend_bits = st.nend_bits;
        ec.nbits_total = st.nbits_total; ec.offs = st.offs; ec.rng = st.rng; ec.val = st.val; ec.ext = st.ext; ec.rem = st.rem; ec.error = st.error;
        int pst = i.ec_prevSignalType, plag = i.ec_prevLagIndex;
        if (i.which & 1) {
            ca::SilkIndices ix;
            for (int k = 0; k < 4; k++) { ix.GainsIndices[k] = i.GainsIndices[k]; ix.LTPIndex[k] = i.LTPIndex[k]; }
            for (int k = 0; k <= 16; k++) ix.NLSFIndices[k] = i.NLSFIndices[k];
            ix.lagIndex = i.lagIndex; ix.contourIndex = i.contourIndex; ix.signalType = i.signalType; ix.quantOffsetType = i.quantOffsetType;
            ix.NLSFInterpCoef_Q2 = i.NLSFInterpCoef_Q2; ix.PERIndex = i.PERIndex; ix.LTP_scaleIndex = i.LTP_scaleIndex; ix.Seed = i.Seed;
            ca::silk_encode_indices_dev(ec, ix, i.nb_subfr, i.fs_kHz, i.predictLPCOrder, i.condCoding, pst, plag);
        }
        if (i.which & 2) {
            uint8_t absq[320];
            ca::silk_encode_pulses_dev(ec, i.signalType, i.quantOffsetType, (const int8_t *)i.pulses, i.frame_length, absq);
        }
        st.end_offs = ec.end_offs; st.end_window = ec.end_window; st.nend_bits = ec.nend_bits; st.nbits_total = ec.nbits_total; st.offs = ec.offs;
        st.rng = ec.rng; st.val = ec.val; st.ext = ec.ext; st.rem = ec.rem; st.error = ec.error;
        out[r].ec_prevSignalType = pst; out[r].ec_prevLagIndex = plag; out[r].status = 0; out[r].reserved = 0;
    }
}

// ---- silk_VAD_GetSA_Q8_c, host build of concentus_amd/csrc/silk_vad_dev.h ----
#include "../../concentus_amd/csrc/silk_vad_dev.h"
extern "C" void emu_silk_vad(const opusgpu_vad_in *in, opusgpu_vad_state *st, opusgpu_vad_out *out, long n)
{
    for (long r = 0; r < n; r++) {
        ca::VadState V;
        for (int k = 0; k < 2; k++) { V.AnaState[k] = st[r].AnaState[k]; V.AnaState1[k] = st[r].AnaState1[k]; V.AnaState2[k] = st[r].AnaState2[k]; }
        for (int k = 0; k < 4; k++) {
            V.XnrgSubfr[k] = st[r].XnrgSubfr[k]; V.NrgRatioSmth_Q8[k] = st[r].NrgRatioSmth_Q8[k]; V.NL[k] = st[r].NL[k]; V.inv_NL[k] = st[r].inv_NL[k];
            V.NoiseLevelBias[k] = st[r].NoiseLevelBias[k];
        }
        V.HPstate = st[r].HPstate; V.counter = st[r].counter;
        ca::VadOut o;
        int16_t X[400];
        ca::silk_VAD_GetSA_Q8_dev(V, o, (const int16_t *)in[r].pIn, (int16_t *)X, in[r].frame_length, in[r].fs_kHz);
        for (int k = 0; k < 2; k++) { st[r].AnaState[k] = V.AnaState[k]; st[r].AnaState1[k] = V.AnaState1[k]; st[r].AnaState2[k] = V.AnaState2[k]; }
        for (int k = 0; k < 4; k++) { st[r].XnrgSubfr[k] = V.XnrgSubfr[k]; st[r].NrgRatioSmth_Q8[k] = V.NrgRatioSmth_Q8[k]; st[r].NL[k] = V.NL[k]; st[r].inv_NL[k] = V.inv_NL[k]; }
        st[r].HPstate = V.HPstate; st[r].counter = V.counter;
        memset(&out[r], 0, sizeof(out[r]));
        out[r].speech_activity_Q8 = o.speech_activity_Q8; out[r].input_tilt_Q15 = o.input_tilt_Q15;
        for (int k = 0; k < 4; k++) out[r].input_quality_bands_Q15[k] = o.input_quality_bands_Q15[k];
    }
}

// ---- the bitrate loop of silk_encode_frame_FIX, host build of concentus_amd/csrc/silk_rate_dev.h: one step per record, nBits given ----
#include "../../concentus_amd/csrc/silk_rate_dev.h"
extern "C" void emu_silk_rate_control(opusgpu_silk_rate_ctl *ctl, const int32_t *nBits, long n)
{
    for (long r = 0; r < n; r++) ca::silk_rate_control_step_dev(ctl[r], nBits[r]);
}

// ---- range coder scripts (csrc/ec_script.h) on the host build: the same device code the hooks opusgpu_ec_enc_script /
// opusgpu_ec_dec_script run, for the CPU tier (tests/test_ec_script_cpu.py). ec = the 11 fields in the order of refcap's ec_pack:
// storage end_offs end_window nend_bits nbits_total offs rng val ext rem error.
#include "../../concentus_amd/csrc/ec_script.h"
extern "C" int emu_ec_enc_script(int32_t *ec, unsigned char *buf, const int32_t *ops, int n)
{
    RangeEnc e;
    e.buf = buf; e.storage = (u32)ec[0]; e.end_offs = (u32)ec[1]; e.end_window = (u32)ec[2]; e.nend_bits = ec[3]; e.nbits_total = ec[4];
    e.offs = (u32)ec[5]; e.rng = (u32)ec[6]; e.val = (u32)ec[7]; e.ext = (u32)ec[8]; e.rem = ec[9]; e.error = ec[10];
    if (!ec_enc_script_ok(ops, n)) return -1;
    ec_enc_run_script(e, ops, n);
    ec[0] = (int32_t)e.storage; ec[1] = (int32_t)e.end_offs; ec[2] = (int32_t)e.end_window; ec[3] = e.nend_bits; ec[4] = e.nbits_total;
    ec[5] = (int32_t)e.offs; ec[6] = (int32_t)e.rng; ec[7] = (int32_t)e.val; ec[8] = (int32_t)e.ext; ec[9] = e.rem; ec[10] = e.error;
    return 0;
}
extern "C" int emu_ec_dec_script(int32_t *ec, const unsigned char *buf, const int32_t *ops, int n, int32_t *out)
{
    RangeDec d;
    d.buf = buf; d.storage = (u32)ec[0]; d.end_offs = (u32)ec[1]; d.end_window = (u32)ec[2]; d.nend_bits = ec[3]; d.nbits_total = ec[4];
    d.offs = (u32)ec[5]; d.rng = (u32)ec[6]; d.val = (u32)ec[7]; d.ext = (u32)ec[8]; d.rem = ec[9]; d.error = ec[10];
    if (!ec_dec_script_ok(ops, n)) return -1;
    ec_dec_run_script(d, ops, n, out);
    ec[1] = (int32_t)d.end_offs; ec[2] = (int32_t)d.end_window; ec[3] = d.nend_bits; ec[4] = d.nbits_total;
    ec[5] = (int32_t)d.offs; ec[6] = (int32_t)d.rng; ec[7] = (int32_t)d.val; ec[8] = (int32_t)d.ext; ec[9] = d.rem; ec[10] = d.error;
    return 0;
}

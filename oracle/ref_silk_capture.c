/* oracle/ref_silk_capture.c -- TEST INFRASTRUCTURE ONLY.
 * Linked with -Wl,--wrap=silk_burg_modified_c,--wrap=silk_NSQ_c,--wrap=silk_NSQ_del_dec_c,--wrap=silk_find_LPC_FIX,--wrap=silk_process_NLSFs,--wrap=silk_residual_energy_FIX,--wrap=silk_find_pred_coefs_FIX,--wrap=silk_process_gains_FIX,--wrap=silk_noise_shape_analysis_FIX,--wrap=silk_prefilter_FIX,--wrap=silk_find_pitch_lags_FIX,--wrap=silk_encode_indices,--wrap=silk_encode_pulses,--wrap=silk_VAD_GetSA_Q8_c,--wrap=silk_encode_frame_FIX into a capture variant of the compiled
 * reference (oracle/_ref/libopus_ref_silkcap.so): every call of the two SILK functions on the hot
 * path is forwarded to the real reference code and its arguments / results are recorded as the flat
 * records of include/opusgpu_silk.h ("capture at the function boundary", SURVEY.md 4: the NailTester
 * technique). Compiled against the reference's own headers where they lie; nothing is copied. */
#include <stdlib.h>
#include <string.h>
#include "main.h"                      /* opus-fix/silk/main.h (via -I) */
#include "../include/opusgpu_silk.h"

typedef char nsq_state_layout_check[sizeof(silk_nsq_state) == sizeof(opusgpu_nsq_state) ? 1 : -1];

/* Frame bookkeeping for the aligned ("chain") capture: g_frame counts calls of silk_find_pitch_lags_FIX (one per encoded frame);
 * every wrap notes the frame its record belongs to, so that the records of one frame can be matched across functions. */
enum { FID_PITCH, FID_SHAPE, FID_FPC, FID_GAINS, FID_PREFILTER, FID_NSQ, FID_DD, FID_BITS_IDX, FID_BITS_PLS, FID_FRAME, FID_KINDS };
static int g_q_calls;      /* quantiser calls (silk_NSQ_c + silk_NSQ_del_dec_c), recorded or not */
static int g_frame, *g_fid[FID_KINDS], g_fid_cap;
static void fid_note(int kind, int rec) { if (g_fid[kind] && rec >= 0 && rec < g_fid_cap) g_fid[kind][rec] = g_frame; }
void refcap_get_frame_ids(int kind, int *out, int n) { if (g_fid[kind]) memcpy(out, g_fid[kind], sizeof(int) * (size_t)n); }

static opusgpu_burg_in *g_bin; static opusgpu_burg_out *g_bout; static int g_nb, g_capb;
static opusgpu_nsq_in *g_nin; static opusgpu_nsq_state *g_nst_in, *g_nst_out; static opusgpu_nsq_out *g_nout;
static int g_nn, g_capn, g_on;

void refcap_start(int max_records)
{
    g_capb = g_capn = max_records; g_nb = g_nn = 0; g_on = 1;
    g_bin = (opusgpu_burg_in *)calloc(max_records, sizeof(*g_bin));
    g_bout = (opusgpu_burg_out *)calloc(max_records, sizeof(*g_bout));
    g_nin = (opusgpu_nsq_in *)calloc(max_records, sizeof(*g_nin));
    g_nst_in = (opusgpu_nsq_state *)calloc(max_records, sizeof(*g_nst_in));
    g_nst_out = (opusgpu_nsq_state *)calloc(max_records, sizeof(*g_nst_out));
    g_nout = (opusgpu_nsq_out *)calloc(max_records, sizeof(*g_nout));
}
int refcap_count_burg(void) { return g_nb; }
int refcap_count_nsq(void) { return g_nn; }
void refcap_get(void *bin, void *bout, void *nin, void *nst_in, void *nst_out, void *nout)
{
    memcpy(bin, g_bin, (size_t)g_nb * sizeof(*g_bin)); memcpy(bout, g_bout, (size_t)g_nb * sizeof(*g_bout));
    memcpy(nin, g_nin, (size_t)g_nn * sizeof(*g_nin)); memcpy(nst_in, g_nst_in, (size_t)g_nn * sizeof(*g_nst_in));
    memcpy(nst_out, g_nst_out, (size_t)g_nn * sizeof(*g_nst_out)); memcpy(nout, g_nout, (size_t)g_nn * sizeof(*g_nout));
}
int refcap_sizes(int which)
{
    switch (which) { case 0: return sizeof(opusgpu_burg_in); case 1: return sizeof(opusgpu_burg_out);
    case 2: return sizeof(opusgpu_nsq_in); case 3: return sizeof(opusgpu_nsq_state); default: return sizeof(opusgpu_nsq_out); }
}

void __real_silk_burg_modified_c(opus_int32 *res_nrg, opus_int *res_nrg_Q, opus_int32 A_Q16[], const opus_int16 x[],
                                 const opus_int32 minInvGain_Q30, const opus_int subfr_length, const opus_int nb_subfr,
                                 const opus_int D, int arch);
void __wrap_silk_burg_modified_c(opus_int32 *res_nrg, opus_int *res_nrg_Q, opus_int32 A_Q16[], const opus_int16 x[],
                                 const opus_int32 minInvGain_Q30, const opus_int subfr_length, const opus_int nb_subfr,
                                 const opus_int D, int arch)
{
    __real_silk_burg_modified_c(res_nrg, res_nrg_Q, A_Q16, x, minInvGain_Q30, subfr_length, nb_subfr, D, arch);
    if (g_on && g_nb < g_capb && subfr_length * nb_subfr <= OPUSGPU_SILK_BURG_MAX_X) {
        opusgpu_burg_in *r = &g_bin[g_nb];
        memcpy(r->x, x, sizeof(opus_int16) * subfr_length * nb_subfr);
        r->minInvGain_Q30 = minInvGain_Q30; r->subfr_length = subfr_length; r->nb_subfr = nb_subfr; r->D = D;
        g_bout[g_nb].res_nrg = *res_nrg; g_bout[g_nb].res_nrg_Q = *res_nrg_Q;
        memcpy(g_bout[g_nb].A_Q16, A_Q16, sizeof(opus_int32) * D);
        g_nb++;
    }
}

void __real_silk_NSQ_c(const silk_encoder_state *psEncC, silk_nsq_state *NSQ, SideInfoIndices *psIndices, const opus_int32 x_Q3[],
                       opus_int8 pulses[], const opus_int16 PredCoef_Q12[], const opus_int16 LTPCoef_Q14[], const opus_int16 AR2_Q13[],
                       const opus_int HarmShapeGain_Q14[], const opus_int Tilt_Q14[], const opus_int32 LF_shp_Q14[],
                       const opus_int32 Gains_Q16[], const opus_int pitchL[], const opus_int Lambda_Q10, const opus_int LTP_scale_Q14);
void __wrap_silk_NSQ_c(const silk_encoder_state *psEncC, silk_nsq_state *NSQ, SideInfoIndices *psIndices, const opus_int32 x_Q3[],
                       opus_int8 pulses[], const opus_int16 PredCoef_Q12[], const opus_int16 LTPCoef_Q14[], const opus_int16 AR2_Q13[],
                       const opus_int HarmShapeGain_Q14[], const opus_int Tilt_Q14[], const opus_int32 LF_shp_Q14[],
                       const opus_int32 Gains_Q16[], const opus_int pitchL[], const opus_int Lambda_Q10, const opus_int LTP_scale_Q14)
{
    int rec = (g_on && g_nin && g_nn < g_capn && psEncC->frame_length <= OPUSGPU_SILK_MAX_FRAME) ? g_nn : -1;
    fid_note(FID_NSQ, rec);
    g_q_calls++;
    if (rec >= 0) {
        opusgpu_nsq_in *r = &g_nin[rec];
        r->nb_subfr = psEncC->nb_subfr; r->subfr_length = psEncC->subfr_length; r->frame_length = psEncC->frame_length;
        r->ltp_mem_length = psEncC->ltp_mem_length; r->predictLPCOrder = psEncC->predictLPCOrder;
        r->shapingLPCOrder = psEncC->shapingLPCOrder;
        r->signalType = psIndices->signalType; r->quantOffsetType = psIndices->quantOffsetType;
        r->NLSFInterpCoef_Q2 = psIndices->NLSFInterpCoef_Q2; r->Seed = psIndices->Seed;
        r->Lambda_Q10 = Lambda_Q10; r->LTP_scale_Q14 = LTP_scale_Q14;
        for (int k = 0; k < 4; k++) {
            r->HarmShapeGain_Q14[k] = HarmShapeGain_Q14[k]; r->Tilt_Q14[k] = Tilt_Q14[k]; r->LF_shp_Q14[k] = LF_shp_Q14[k];
            r->Gains_Q16[k] = Gains_Q16[k]; r->pitchL[k] = pitchL[k];
        }
        memcpy(r->x_Q3, x_Q3, sizeof(opus_int32) * psEncC->frame_length);
        memcpy(r->PredCoef_Q12, PredCoef_Q12, sizeof(r->PredCoef_Q12));
        memcpy(r->LTPCoef_Q14, LTPCoef_Q14, sizeof(r->LTPCoef_Q14));
        memcpy(r->AR2_Q13, AR2_Q13, sizeof(r->AR2_Q13));
        memcpy(&g_nst_in[rec], NSQ, sizeof(*NSQ));
    }
    __real_silk_NSQ_c(psEncC, NSQ, psIndices, x_Q3, pulses, PredCoef_Q12, LTPCoef_Q14, AR2_Q13, HarmShapeGain_Q14, Tilt_Q14,
                      LF_shp_Q14, Gains_Q16, pitchL, Lambda_Q10, LTP_scale_Q14);
    if (rec >= 0) {
        memcpy(&g_nst_out[rec], NSQ, sizeof(*NSQ));
        memcpy(g_nout[rec].pulses, pulses, psEncC->frame_length);
        g_nn++;
    }
}

/* ---- silk_NSQ_del_dec_c (opus-fix/silk/NSQ_del_dec.c:112) ---- */
static opusgpu_nsq_dd_in *g_din; static opusgpu_nsq_state *g_dst_in, *g_dst_out; static opusgpu_nsq_dd_out *g_dout;
static int g_nd, g_capd;
void refcap_start_dd(int max_records)
{
    g_capd = max_records; g_nd = 0; g_on = 1;
    g_din = (opusgpu_nsq_dd_in *)calloc(max_records, sizeof(*g_din));
    g_dst_in = (opusgpu_nsq_state *)calloc(max_records, sizeof(*g_dst_in));
    g_dst_out = (opusgpu_nsq_state *)calloc(max_records, sizeof(*g_dst_out));
    g_dout = (opusgpu_nsq_dd_out *)calloc(max_records, sizeof(*g_dout));
}
int refcap_count_dd(void) { return g_nd; }
int refcap_sizes_dd(int which) { return which == 0 ? sizeof(opusgpu_nsq_dd_in) : which == 1 ? sizeof(opusgpu_nsq_state) : sizeof(opusgpu_nsq_dd_out); }
void refcap_get_dd(void *din, void *st_in, void *st_out, void *dout)
{
    memcpy(din, g_din, (size_t)g_nd * sizeof(*g_din)); memcpy(st_in, g_dst_in, (size_t)g_nd * sizeof(*g_dst_in));
    memcpy(st_out, g_dst_out, (size_t)g_nd * sizeof(*g_dst_out)); memcpy(dout, g_dout, (size_t)g_nd * sizeof(*g_dout));
}

void __real_silk_NSQ_del_dec_c(const silk_encoder_state *psEncC, silk_nsq_state *NSQ, SideInfoIndices *psIndices, const opus_int32 x_Q3[],
                       opus_int8 pulses[], const opus_int16 PredCoef_Q12[], const opus_int16 LTPCoef_Q14[], const opus_int16 AR2_Q13[],
                       const opus_int HarmShapeGain_Q14[], const opus_int Tilt_Q14[], const opus_int32 LF_shp_Q14[],
                       const opus_int32 Gains_Q16[], const opus_int pitchL[], const opus_int Lambda_Q10, const opus_int LTP_scale_Q14);
void __wrap_silk_NSQ_del_dec_c(const silk_encoder_state *psEncC, silk_nsq_state *NSQ, SideInfoIndices *psIndices, const opus_int32 x_Q3[],
                       opus_int8 pulses[], const opus_int16 PredCoef_Q12[], const opus_int16 LTPCoef_Q14[], const opus_int16 AR2_Q13[],
                       const opus_int HarmShapeGain_Q14[], const opus_int Tilt_Q14[], const opus_int32 LF_shp_Q14[],
                       const opus_int32 Gains_Q16[], const opus_int pitchL[], const opus_int Lambda_Q10, const opus_int LTP_scale_Q14)
{
    int rec = (g_on && g_din && g_nd < g_capd && psEncC->frame_length <= OPUSGPU_SILK_MAX_FRAME) ? g_nd : -1;
    fid_note(FID_DD, rec);
    g_q_calls++;
    if (rec >= 0) {
        opusgpu_nsq_in *r = &g_din[rec].base;
        r->nb_subfr = psEncC->nb_subfr; r->subfr_length = psEncC->subfr_length; r->frame_length = psEncC->frame_length;
        r->ltp_mem_length = psEncC->ltp_mem_length; r->predictLPCOrder = psEncC->predictLPCOrder;
        r->shapingLPCOrder = psEncC->shapingLPCOrder;
        r->signalType = psIndices->signalType; r->quantOffsetType = psIndices->quantOffsetType;
        r->NLSFInterpCoef_Q2 = psIndices->NLSFInterpCoef_Q2; r->Seed = psIndices->Seed;
        r->Lambda_Q10 = Lambda_Q10; r->LTP_scale_Q14 = LTP_scale_Q14;
        for (int k = 0; k < 4; k++) {
            r->HarmShapeGain_Q14[k] = HarmShapeGain_Q14[k]; r->Tilt_Q14[k] = Tilt_Q14[k]; r->LF_shp_Q14[k] = LF_shp_Q14[k];
            r->Gains_Q16[k] = Gains_Q16[k]; r->pitchL[k] = pitchL[k];
        }
        memcpy(r->x_Q3, x_Q3, sizeof(opus_int32) * psEncC->frame_length);
        memcpy(r->PredCoef_Q12, PredCoef_Q12, sizeof(r->PredCoef_Q12));
        memcpy(r->LTPCoef_Q14, LTPCoef_Q14, sizeof(r->LTPCoef_Q14));
        memcpy(r->AR2_Q13, AR2_Q13, sizeof(r->AR2_Q13));
        g_din[rec].nStatesDelayedDecision = psEncC->nStatesDelayedDecision;
        g_din[rec].warping_Q16 = psEncC->warping_Q16;
        memcpy(&g_dst_in[rec], NSQ, sizeof(*NSQ));
    }
    __real_silk_NSQ_del_dec_c(psEncC, NSQ, psIndices, x_Q3, pulses, PredCoef_Q12, LTPCoef_Q14, AR2_Q13, HarmShapeGain_Q14, Tilt_Q14,
                      LF_shp_Q14, Gains_Q16, pitchL, Lambda_Q10, LTP_scale_Q14);
    if (rec >= 0) {
        memcpy(&g_dst_out[rec], NSQ, sizeof(*NSQ));
        memcpy(g_dout[rec].pulses, pulses, psEncC->frame_length);
        g_dout[rec].Seed = psIndices->Seed;
        g_nd++;
    }
}

/* ---- silk_find_LPC_FIX (opus-fix/silk/fixed/find_LPC_FIX.c:37): arguments + the psEncC fields it reads -> NLSF_Q15, NLSFInterpCoef_Q2 ---- */
#include "main_FIX.h"
static opusgpu_find_lpc_in *g_lin; static opusgpu_find_lpc_out *g_lout; static int g_nl, g_capl;
void refcap_start_lpc(int max_records)
{
    g_capl = max_records; g_nl = 0; g_on = 1;
    g_lin = (opusgpu_find_lpc_in *)calloc(max_records, sizeof(*g_lin));
    g_lout = (opusgpu_find_lpc_out *)calloc(max_records, sizeof(*g_lout));
}
int refcap_count_lpc(void) { return g_nl; }
int refcap_sizes_lpc(int which) { return which == 0 ? sizeof(opusgpu_find_lpc_in) : sizeof(opusgpu_find_lpc_out); }
void refcap_get_lpc(void *lin, void *lout)
{
    memcpy(lin, g_lin, (size_t)g_nl * sizeof(*g_lin)); memcpy(lout, g_lout, (size_t)g_nl * sizeof(*g_lout));
}

void __real_silk_find_LPC_FIX(silk_encoder_state *psEncC, opus_int16 NLSF_Q15[], const opus_int16 x[], const opus_int32 minInvGain_Q30);
void __wrap_silk_find_LPC_FIX(silk_encoder_state *psEncC, opus_int16 NLSF_Q15[], const opus_int16 x[], const opus_int32 minInvGain_Q30)
{
    const int nx = (psEncC->subfr_length + psEncC->predictLPCOrder) * psEncC->nb_subfr;
    int rec = (g_on && g_lin && g_nl < g_capl && nx <= OPUSGPU_SILK_BURG_MAX_X) ? g_nl : -1;
    if (rec >= 0) {
        opusgpu_find_lpc_in *r = &g_lin[rec];
        memcpy(r->x, x, sizeof(opus_int16) * nx);
        r->minInvGain_Q30 = minInvGain_Q30; r->subfr_length = psEncC->subfr_length; r->nb_subfr = psEncC->nb_subfr;
        r->predictLPCOrder = psEncC->predictLPCOrder; r->useInterpolatedNLSFs = psEncC->useInterpolatedNLSFs;
        r->first_frame_after_reset = psEncC->first_frame_after_reset;
        memcpy(r->prev_NLSFq_Q15, psEncC->prev_NLSFq_Q15, sizeof(r->prev_NLSFq_Q15));
    }
    __real_silk_find_LPC_FIX(psEncC, NLSF_Q15, x, minInvGain_Q30);
    if (rec >= 0) {
        memcpy(g_lout[rec].NLSF_Q15, NLSF_Q15, sizeof(opus_int16) * psEncC->predictLPCOrder);
        g_lout[rec].NLSFInterpCoef_Q2 = psEncC->indices.NLSFInterpCoef_Q2;
        g_lout[rec].status = 0;
        g_nl++;
    }
}

/* ---- silk_process_NLSFs (opus-fix/silk/process_NLSFs.c:35) and silk_residual_energy_FIX (opus-fix/silk/fixed/residual_energy_FIX.c:37):
 * the two calls silk_find_pred_coefs_FIX makes after silk_find_LPC_FIX (find_pred_coefs_FIX.c:139-143) ---- */
static opusgpu_process_nlsf_in *g_pin; static opusgpu_process_nlsf_out *g_pout; static int g_np;
static opusgpu_res_nrg_in *g_ein; static opusgpu_res_nrg_out *g_eout; static int g_ne, g_capp;
void refcap_start_pred(int max_records)
{
    g_capp = max_records; g_np = g_ne = 0; g_on = 1;
    g_pin = (opusgpu_process_nlsf_in *)calloc(max_records, sizeof(*g_pin));
    g_pout = (opusgpu_process_nlsf_out *)calloc(max_records, sizeof(*g_pout));
    g_ein = (opusgpu_res_nrg_in *)calloc(max_records, sizeof(*g_ein));
    g_eout = (opusgpu_res_nrg_out *)calloc(max_records, sizeof(*g_eout));
}
int refcap_count_pred(int which) { return which == 0 ? g_np : g_ne; }
int refcap_sizes_pred(int which)
{
    return which == 0 ? sizeof(opusgpu_process_nlsf_in) : which == 1 ? sizeof(opusgpu_process_nlsf_out)
         : which == 2 ? sizeof(opusgpu_res_nrg_in) : sizeof(opusgpu_res_nrg_out);
}
void refcap_get_pred(void *pin, void *pout, void *ein, void *eout)
{
    memcpy(pin, g_pin, (size_t)g_np * sizeof(*g_pin)); memcpy(pout, g_pout, (size_t)g_np * sizeof(*g_pout));
    memcpy(ein, g_ein, (size_t)g_ne * sizeof(*g_ein)); memcpy(eout, g_eout, (size_t)g_ne * sizeof(*g_eout));
}

void __real_silk_process_NLSFs(silk_encoder_state *psEncC, opus_int16 PredCoef_Q12[2][MAX_LPC_ORDER], opus_int16 pNLSF_Q15[MAX_LPC_ORDER],
                               const opus_int16 prev_NLSFq_Q15[MAX_LPC_ORDER]);
void __wrap_silk_process_NLSFs(silk_encoder_state *psEncC, opus_int16 PredCoef_Q12[2][MAX_LPC_ORDER], opus_int16 pNLSF_Q15[MAX_LPC_ORDER],
                               const opus_int16 prev_NLSFq_Q15[MAX_LPC_ORDER])
{
    int rec = (g_on && g_pin && g_np < g_capp) ? g_np : -1;
    if (rec >= 0) {
        opusgpu_process_nlsf_in *r = &g_pin[rec];
        memcpy(r->NLSF_Q15, pNLSF_Q15, sizeof(opus_int16) * psEncC->predictLPCOrder);
        memcpy(r->prev_NLSFq_Q15, prev_NLSFq_Q15, sizeof(opus_int16) * psEncC->predictLPCOrder);
        r->speech_activity_Q8 = psEncC->speech_activity_Q8; r->nb_subfr = psEncC->nb_subfr; r->predictLPCOrder = psEncC->predictLPCOrder;
        r->useInterpolatedNLSFs = psEncC->useInterpolatedNLSFs; r->NLSFInterpCoef_Q2 = psEncC->indices.NLSFInterpCoef_Q2;
        r->NLSF_MSVQ_Survivors = psEncC->NLSF_MSVQ_Survivors; r->signalType = psEncC->indices.signalType;
    }
    __real_silk_process_NLSFs(psEncC, PredCoef_Q12, pNLSF_Q15, prev_NLSFq_Q15);
    if (rec >= 0) {
        opusgpu_process_nlsf_out *o = &g_pout[rec];
        memcpy(o->PredCoef_Q12[0], PredCoef_Q12[0], sizeof(opus_int16) * psEncC->predictLPCOrder);
        memcpy(o->PredCoef_Q12[1], PredCoef_Q12[1], sizeof(opus_int16) * psEncC->predictLPCOrder);
        memcpy(o->NLSF_Q15, pNLSF_Q15, sizeof(opus_int16) * psEncC->predictLPCOrder);
        memcpy(o->NLSFIndices, psEncC->indices.NLSFIndices, psEncC->predictLPCOrder + 1);
        o->status = 0;
        g_np++;
    }
}

void __real_silk_residual_energy_FIX(opus_int32 nrgs[MAX_NB_SUBFR], opus_int nrgsQ[MAX_NB_SUBFR], const opus_int16 x[],
                                     opus_int16 a_Q12[2][MAX_LPC_ORDER], const opus_int32 gains[MAX_NB_SUBFR], const opus_int subfr_length,
                                     const opus_int nb_subfr, const opus_int LPC_order, int arch);
void __wrap_silk_residual_energy_FIX(opus_int32 nrgs[MAX_NB_SUBFR], opus_int nrgsQ[MAX_NB_SUBFR], const opus_int16 x[],
                                     opus_int16 a_Q12[2][MAX_LPC_ORDER], const opus_int32 gains[MAX_NB_SUBFR], const opus_int subfr_length,
                                     const opus_int nb_subfr, const opus_int LPC_order, int arch)
{
    const int nx = (subfr_length + LPC_order) * nb_subfr;
    int rec = (g_on && g_ein && g_ne < g_capp && nx <= OPUSGPU_SILK_BURG_MAX_X) ? g_ne : -1;
    if (rec >= 0) {
        opusgpu_res_nrg_in *r = &g_ein[rec];
        memcpy(r->x, x, sizeof(opus_int16) * nx);
        memcpy(r->a_Q12[0], a_Q12[0], sizeof(opus_int16) * LPC_order); memcpy(r->a_Q12[1], a_Q12[1], sizeof(opus_int16) * LPC_order);
        memcpy(r->gains, gains, sizeof(opus_int32) * nb_subfr);
        r->subfr_length = subfr_length; r->nb_subfr = nb_subfr; r->LPC_order = LPC_order;
    }
    __real_silk_residual_energy_FIX(nrgs, nrgsQ, x, a_Q12, gains, subfr_length, nb_subfr, LPC_order, arch);
    if (rec >= 0) {
        memcpy(g_eout[rec].nrgs, nrgs, sizeof(opus_int32) * nb_subfr);
        for (int k = 0; k < nb_subfr; k++) g_eout[rec].nrgsQ[k] = nrgsQ[k];
        g_eout[rec].status = 0;
        g_ne++;
    }
}

/* ---- silk_find_pred_coefs_FIX (opus-fix/silk/fixed/find_pred_coefs_FIX.c:35), whole: arguments + the psEnc / psEncCtrl fields it
 * reads -> every field it writes ---- */
static opusgpu_find_pred_coefs_in *g_fin; static opusgpu_find_pred_coefs_out *g_fout; static int g_nf, g_capf;
void refcap_start_fpc(int max_records)
{
    g_capf = max_records; g_nf = 0; g_on = 1;
    g_fin = (opusgpu_find_pred_coefs_in *)calloc(max_records, sizeof(*g_fin));
    g_fout = (opusgpu_find_pred_coefs_out *)calloc(max_records, sizeof(*g_fout));
}
int refcap_count_fpc(void) { return g_nf; }
int refcap_sizes_fpc(int which) { return which == 0 ? sizeof(opusgpu_find_pred_coefs_in) : sizeof(opusgpu_find_pred_coefs_out); }
void refcap_get_fpc(void *fin, void *fout)
{
    memcpy(fin, g_fin, (size_t)g_nf * sizeof(*g_fin)); memcpy(fout, g_fout, (size_t)g_nf * sizeof(*g_fout));
}

void __real_silk_find_pred_coefs_FIX(silk_encoder_state_FIX *psEnc, silk_encoder_control_FIX *psEncCtrl, const opus_int16 res_pitch[],
                                     const opus_int16 x[], opus_int condCoding);
void __wrap_silk_find_pred_coefs_FIX(silk_encoder_state_FIX *psEnc, silk_encoder_control_FIX *psEncCtrl, const opus_int16 res_pitch[],
                                     const opus_int16 x[], opus_int condCoding)
{
    const silk_encoder_state *c = &psEnc->sCmn;
    int rec = (g_on && g_fin && g_nf < g_capf && c->ltp_mem_length <= OPUSGPU_SILK_MAX_LTP_MEM && c->frame_length <= OPUSGPU_SILK_MAX_FRAME)
                  ? g_nf : -1;
    fid_note(FID_FPC, rec);
    if (rec >= 0) {
        opusgpu_find_pred_coefs_in *r = &g_fin[rec];
        memcpy(r->res_pitch, res_pitch, sizeof(opus_int16) * (c->ltp_mem_length + c->frame_length));
        memcpy(r->x, x - c->ltp_mem_length, sizeof(opus_int16) * (c->ltp_mem_length + c->frame_length));
        for (int k = 0; k < MAX_NB_SUBFR; k++) { r->Gains_Q16[k] = psEncCtrl->Gains_Q16[k]; r->pitchL[k] = psEncCtrl->pitchL[k]; }
        memcpy(r->prev_NLSFq_Q15, c->prev_NLSFq_Q15, sizeof(r->prev_NLSFq_Q15));
        r->nb_subfr = c->nb_subfr; r->subfr_length = c->subfr_length; r->predictLPCOrder = c->predictLPCOrder; r->ltp_mem_length = c->ltp_mem_length;
        r->signalType = c->indices.signalType; r->condCoding = condCoding; r->first_frame_after_reset = c->first_frame_after_reset;
        r->useInterpolatedNLSFs = c->useInterpolatedNLSFs; r->speech_activity_Q8 = c->speech_activity_Q8;
        r->NLSF_MSVQ_Survivors = c->NLSF_MSVQ_Survivors; r->mu_LTP_Q9 = c->mu_LTP_Q9; r->LTPQuantLowComplexity = c->LTPQuantLowComplexity;
        r->sum_log_gain_Q7 = c->sum_log_gain_Q7; r->coding_quality_Q14 = psEncCtrl->coding_quality_Q14;
        r->PacketLoss_perc = c->PacketLoss_perc; r->nFramesPerPacket = c->nFramesPerPacket;
    }
    __real_silk_find_pred_coefs_FIX(psEnc, psEncCtrl, res_pitch, x, condCoding);
    if (rec >= 0) {
        opusgpu_find_pred_coefs_out *o = &g_fout[rec];
        const int voiced = c->indices.signalType == TYPE_VOICED;
        memcpy(o->PredCoef_Q12[0], psEncCtrl->PredCoef_Q12[0], sizeof(opus_int16) * c->predictLPCOrder);
        memcpy(o->PredCoef_Q12[1], psEncCtrl->PredCoef_Q12[1], sizeof(opus_int16) * c->predictLPCOrder);
        memcpy(o->LTPCoef_Q14, psEncCtrl->LTPCoef_Q14, sizeof(opus_int16) * c->nb_subfr * LTP_ORDER);
        memcpy(o->NLSF_Q15, c->prev_NLSFq_Q15, sizeof(opus_int16) * c->predictLPCOrder);
        for (int k = 0; k < c->nb_subfr; k++) { o->ResNrg[k] = psEncCtrl->ResNrg[k]; o->ResNrgQ[k] = psEncCtrl->ResNrgQ[k]; }
        o->LTPredCodGain_Q7 = psEncCtrl->LTPredCodGain_Q7; o->sum_log_gain_Q7 = c->sum_log_gain_Q7;
        o->LTP_scale_Q14 = voiced ? psEncCtrl->LTP_scale_Q14 : 0;
        memcpy(o->NLSFIndices, c->indices.NLSFIndices, c->predictLPCOrder + 1);
        o->NLSFInterpCoef_Q2 = c->indices.NLSFInterpCoef_Q2;
        if (voiced) { memcpy(o->LTPIndex, c->indices.LTPIndex, c->nb_subfr); o->PERIndex = c->indices.PERIndex; }
        o->LTP_scaleIndex = voiced ? c->indices.LTP_scaleIndex : -1;
        o->status = 0;
        g_nf++;
    }
}

/* ---- silk_process_gains_FIX (opus-fix/silk/fixed/process_gains_FIX.c:37): the psEnc / psEncCtrl fields it reads -> the ones it writes ---- */
static opusgpu_process_gains_in *g_gin; static opusgpu_process_gains_out *g_gout; static int g_ng, g_capg;
void refcap_start_gains(int max_records)
{
    g_capg = max_records; g_ng = 0; g_on = 1;
    g_gin = (opusgpu_process_gains_in *)calloc(max_records, sizeof(*g_gin));
    g_gout = (opusgpu_process_gains_out *)calloc(max_records, sizeof(*g_gout));
}
int refcap_count_gains(void) { return g_ng; }
int refcap_sizes_gains(int which) { return which == 0 ? sizeof(opusgpu_process_gains_in) : sizeof(opusgpu_process_gains_out); }
void refcap_get_gains(void *gin, void *gout)
{
    memcpy(gin, g_gin, (size_t)g_ng * sizeof(*g_gin)); memcpy(gout, g_gout, (size_t)g_ng * sizeof(*g_gout));
}

void __real_silk_process_gains_FIX(silk_encoder_state_FIX *psEnc, silk_encoder_control_FIX *psEncCtrl, opus_int condCoding);
void __wrap_silk_process_gains_FIX(silk_encoder_state_FIX *psEnc, silk_encoder_control_FIX *psEncCtrl, opus_int condCoding)
{
    const silk_encoder_state *c = &psEnc->sCmn;
    int rec = (g_on && g_gin && g_ng < g_capg) ? g_ng : -1;
    fid_note(FID_GAINS, rec);
    if (rec >= 0) {
        opusgpu_process_gains_in *r = &g_gin[rec];
        for (int k = 0; k < MAX_NB_SUBFR; k++) { r->Gains_Q16[k] = psEncCtrl->Gains_Q16[k]; r->ResNrg[k] = psEncCtrl->ResNrg[k]; r->ResNrgQ[k] = psEncCtrl->ResNrgQ[k]; }
        r->LTPredCodGain_Q7 = psEncCtrl->LTPredCodGain_Q7; r->signalType = c->indices.signalType; r->nb_subfr = c->nb_subfr;
        r->subfr_length = c->subfr_length; r->SNR_dB_Q7 = c->SNR_dB_Q7; r->LastGainIndex = psEnc->sShape.LastGainIndex;
        r->condCoding = condCoding; r->input_tilt_Q15 = c->input_tilt_Q15; r->quantOffsetType = c->indices.quantOffsetType;
        r->nStatesDelayedDecision = c->nStatesDelayedDecision; r->speech_activity_Q8 = c->speech_activity_Q8;
        r->input_quality_Q14 = psEncCtrl->input_quality_Q14; r->coding_quality_Q14 = psEncCtrl->coding_quality_Q14;
    }
    __real_silk_process_gains_FIX(psEnc, psEncCtrl, condCoding);
    if (rec >= 0) {
        opusgpu_process_gains_out *o = &g_gout[rec];
        for (int k = 0; k < c->nb_subfr; k++) {
            o->Gains_Q16[k] = psEncCtrl->Gains_Q16[k]; o->GainsUnq_Q16[k] = psEncCtrl->GainsUnq_Q16[k]; o->GainsIndices[k] = c->indices.GainsIndices[k];
        }
        o->Lambda_Q10 = psEncCtrl->Lambda_Q10; o->LastGainIndex = psEnc->sShape.LastGainIndex; o->lastGainIndexPrev = psEncCtrl->lastGainIndexPrev;
        o->quantOffsetType = c->indices.quantOffsetType; o->status = 0;
        g_ng++;
    }
}

/* ---- silk_noise_shape_analysis_FIX (opus-fix/silk/fixed/noise_shape_analysis_FIX.c:146): arguments + the psEnc / psEncCtrl fields it
 * reads -> every field it writes ---- */
static opusgpu_noise_shape_in *g_sin; static opusgpu_noise_shape_out *g_sout; static int g_ns, g_caps;
void refcap_start_shape(int max_records)
{
    g_caps = max_records; g_ns = 0; g_on = 1;
    g_sin = (opusgpu_noise_shape_in *)calloc(max_records, sizeof(*g_sin));
    g_sout = (opusgpu_noise_shape_out *)calloc(max_records, sizeof(*g_sout));
}
int refcap_count_shape(void) { return g_ns; }
int refcap_sizes_shape(int which) { return which == 0 ? sizeof(opusgpu_noise_shape_in) : sizeof(opusgpu_noise_shape_out); }
void refcap_get_shape(void *sin_, void *sout)
{
    memcpy(sin_, g_sin, (size_t)g_ns * sizeof(*g_sin)); memcpy(sout, g_sout, (size_t)g_ns * sizeof(*g_sout));
}

void __real_silk_noise_shape_analysis_FIX(silk_encoder_state_FIX *psEnc, silk_encoder_control_FIX *psEncCtrl, const opus_int16 *pitch_res,
                                          const opus_int16 *x, int arch);
void __wrap_silk_noise_shape_analysis_FIX(silk_encoder_state_FIX *psEnc, silk_encoder_control_FIX *psEncCtrl, const opus_int16 *pitch_res,
                                          const opus_int16 *x, int arch)
{
    const silk_encoder_state *c = &psEnc->sCmn;
    int rec = (g_on && g_sin && g_ns < g_caps && c->la_shape <= OPUSGPU_SILK_MAX_LA_SHAPE && c->frame_length <= OPUSGPU_SILK_MAX_FRAME) ? g_ns : -1;
    fid_note(FID_SHAPE, rec);
    if (rec >= 0) {
        opusgpu_noise_shape_in *r = &g_sin[rec];
        memcpy(r->x, x - c->la_shape, sizeof(opus_int16) * (c->frame_length + 2 * c->la_shape));
        memcpy(r->pitch_res, pitch_res, sizeof(opus_int16) * c->frame_length);
        r->fs_kHz = c->fs_kHz; r->nb_subfr = c->nb_subfr; r->subfr_length = c->subfr_length; r->la_shape = c->la_shape;
        r->shapeWinLength = c->shapeWinLength; r->shapingLPCOrder = c->shapingLPCOrder; r->warping_Q16 = c->warping_Q16;
        r->SNR_dB_Q7 = c->SNR_dB_Q7; r->useCBR = c->useCBR; r->speech_activity_Q8 = c->speech_activity_Q8;
        r->signalType = c->indices.signalType; r->LTPCorr_Q15 = psEnc->LTPCorr_Q15;
        r->input_quality_bands_Q15[0] = c->input_quality_bands_Q15[0]; r->input_quality_bands_Q15[1] = c->input_quality_bands_Q15[1];
        r->predGain_Q16 = psEncCtrl->predGain_Q16;
        for (int k = 0; k < MAX_NB_SUBFR; k++) r->pitchL[k] = psEncCtrl->pitchL[k];
        r->HarmBoost_smth_Q16 = psEnc->sShape.HarmBoost_smth_Q16; r->HarmShapeGain_smth_Q16 = psEnc->sShape.HarmShapeGain_smth_Q16;
        r->Tilt_smth_Q16 = psEnc->sShape.Tilt_smth_Q16;
    }
    __real_silk_noise_shape_analysis_FIX(psEnc, psEncCtrl, pitch_res, x, arch);
    if (rec >= 0) {
        opusgpu_noise_shape_out *o = &g_sout[rec];
        for (int k = 0; k < c->nb_subfr; k++) {
            o->Gains_Q16[k] = psEncCtrl->Gains_Q16[k]; o->GainsPre_Q14[k] = psEncCtrl->GainsPre_Q14[k]; o->LF_shp_Q14[k] = psEncCtrl->LF_shp_Q14[k];
            memcpy(&o->AR1_Q13[k * 16], &psEncCtrl->AR1_Q13[k * MAX_SHAPE_LPC_ORDER], sizeof(opus_int16) * c->shapingLPCOrder);
            memcpy(&o->AR2_Q13[k * 16], &psEncCtrl->AR2_Q13[k * MAX_SHAPE_LPC_ORDER], sizeof(opus_int16) * c->shapingLPCOrder);
        }
        for (int k = 0; k < MAX_NB_SUBFR; k++) {
            o->HarmBoost_Q14[k] = psEncCtrl->HarmBoost_Q14[k]; o->HarmShapeGain_Q14[k] = psEncCtrl->HarmShapeGain_Q14[k]; o->Tilt_Q14[k] = psEncCtrl->Tilt_Q14[k];
        }
        o->HarmBoost_smth_Q16 = psEnc->sShape.HarmBoost_smth_Q16; o->HarmShapeGain_smth_Q16 = psEnc->sShape.HarmShapeGain_smth_Q16;
        o->Tilt_smth_Q16 = psEnc->sShape.Tilt_smth_Q16;
        o->input_quality_Q14 = psEncCtrl->input_quality_Q14; o->coding_quality_Q14 = psEncCtrl->coding_quality_Q14;
        o->sparseness_Q8 = psEncCtrl->sparseness_Q8; o->quantOffsetType = c->indices.quantOffsetType; o->status = 0;
        g_ns++;
    }
}

/* ---- silk_prefilter_FIX (opus-fix/silk/fixed/prefilter_FIX.c:102): arguments + the fields read, psEnc->sPrefilt before / after -> xw_Q3 ---- */
static opusgpu_prefilter_in *g_xin; static opusgpu_prefilter_state *g_xst0, *g_xst1; static opusgpu_prefilter_out *g_xout; static int g_nx, g_capx;
void refcap_start_prefilter(int max_records)
{
    g_capx = max_records; g_nx = 0; g_on = 1;
    g_xin = (opusgpu_prefilter_in *)calloc(max_records, sizeof(*g_xin));
    g_xst0 = (opusgpu_prefilter_state *)calloc(max_records, sizeof(*g_xst0));
    g_xst1 = (opusgpu_prefilter_state *)calloc(max_records, sizeof(*g_xst1));
    g_xout = (opusgpu_prefilter_out *)calloc(max_records, sizeof(*g_xout));
}
int refcap_count_prefilter(void) { return g_nx; }
int refcap_sizes_prefilter(int which)
{
    return which == 0 ? sizeof(opusgpu_prefilter_in) : which == 1 ? sizeof(opusgpu_prefilter_state) : which == 2 ? sizeof(opusgpu_prefilter_out)
         : (int)sizeof(silk_prefilter_state_FIX);
}
void refcap_get_prefilter(void *xin, void *st0, void *st1, void *xout)
{
    memcpy(xin, g_xin, (size_t)g_nx * sizeof(*g_xin)); memcpy(st0, g_xst0, (size_t)g_nx * sizeof(*g_xst0));
    memcpy(st1, g_xst1, (size_t)g_nx * sizeof(*g_xst1)); memcpy(xout, g_xout, (size_t)g_nx * sizeof(*g_xout));
}

void __real_silk_prefilter_FIX(silk_encoder_state_FIX *psEnc, const silk_encoder_control_FIX *psEncCtrl, opus_int32 xw_Q3[], const opus_int16 x[]);
void __wrap_silk_prefilter_FIX(silk_encoder_state_FIX *psEnc, const silk_encoder_control_FIX *psEncCtrl, opus_int32 xw_Q3[], const opus_int16 x[])
{
    typedef char prefilter_state_layout[sizeof(opusgpu_prefilter_state) == sizeof(silk_prefilter_state_FIX) ? 1 : -1];
    const silk_encoder_state *c = &psEnc->sCmn;
    int rec = (g_on && g_xin && g_nx < g_capx && c->frame_length <= OPUSGPU_SILK_MAX_FRAME) ? g_nx : -1;
    fid_note(FID_PREFILTER, rec);
    if (rec >= 0) {
        opusgpu_prefilter_in *r = &g_xin[rec];
        memcpy(r->x, x, sizeof(opus_int16) * c->frame_length);
        memcpy(r->AR1_Q13, psEncCtrl->AR1_Q13, sizeof(r->AR1_Q13));
        for (int k = 0; k < MAX_NB_SUBFR; k++) {
            r->pitchL[k] = psEncCtrl->pitchL[k]; r->HarmShapeGain_Q14[k] = psEncCtrl->HarmShapeGain_Q14[k]; r->HarmBoost_Q14[k] = psEncCtrl->HarmBoost_Q14[k];
            r->Tilt_Q14[k] = psEncCtrl->Tilt_Q14[k]; r->GainsPre_Q14[k] = psEncCtrl->GainsPre_Q14[k]; r->LF_shp_Q14[k] = psEncCtrl->LF_shp_Q14[k];
        }
        r->coding_quality_Q14 = psEncCtrl->coding_quality_Q14; r->nb_subfr = c->nb_subfr; r->subfr_length = c->subfr_length;
        r->signalType = c->indices.signalType; r->warping_Q16 = c->warping_Q16; r->shapingLPCOrder = c->shapingLPCOrder;
        memcpy(&g_xst0[rec], &psEnc->sPrefilt, sizeof(g_xst0[rec]));
    }
    __real_silk_prefilter_FIX(psEnc, psEncCtrl, xw_Q3, x);
    if (rec >= 0) {
        memcpy(&g_xst1[rec], &psEnc->sPrefilt, sizeof(g_xst1[rec]));
        memcpy(g_xout[rec].xw_Q3, xw_Q3, sizeof(opus_int32) * c->frame_length);
        g_xout[rec].status = 0;
        g_nx++;
    }
}

/* ---- silk_find_pitch_lags_FIX (opus-fix/silk/fixed/find_pitch_lags_FIX.c:37): arguments + the psEnc fields it reads -> res[], pitch lags,
 * indices, voicing, LTPCorr_Q15, predGain_Q16 ---- */
static opusgpu_find_pitch_lags_in *g_tin; static opusgpu_find_pitch_lags_out *g_tout; static int g_nt, g_capt;
void refcap_start_pitch(int max_records)
{
    g_capt = max_records; g_nt = 0; g_on = 1;
    g_tin = (opusgpu_find_pitch_lags_in *)calloc(max_records, sizeof(*g_tin));
    g_tout = (opusgpu_find_pitch_lags_out *)calloc(max_records, sizeof(*g_tout));
}
int refcap_count_pitch(void) { return g_nt; }
int refcap_sizes_pitch(int which) { return which == 0 ? sizeof(opusgpu_find_pitch_lags_in) : sizeof(opusgpu_find_pitch_lags_out); }
void refcap_get_pitch(void *tin, void *tout)
{
    memcpy(tin, g_tin, (size_t)g_nt * sizeof(*g_tin)); memcpy(tout, g_tout, (size_t)g_nt * sizeof(*g_tout));
}

void __real_silk_find_pitch_lags_FIX(silk_encoder_state_FIX *psEnc, silk_encoder_control_FIX *psEncCtrl, opus_int16 res[], const opus_int16 x[], int arch);
void __wrap_silk_find_pitch_lags_FIX(silk_encoder_state_FIX *psEnc, silk_encoder_control_FIX *psEncCtrl, opus_int16 res[], const opus_int16 x[], int arch)
{
    const silk_encoder_state *c = &psEnc->sCmn;
    const int buf_len = c->la_pitch + c->frame_length + c->ltp_mem_length;
    g_frame++;
    int rec = (g_on && g_tin && g_nt < g_capt && buf_len <= OPUSGPU_SILK_PITCH_BUF) ? g_nt : -1;
    fid_note(FID_PITCH, rec);
    if (rec >= 0) {
        opusgpu_find_pitch_lags_in *r = &g_tin[rec];
        memcpy(r->x_buf, x - c->ltp_mem_length, sizeof(opus_int16) * buf_len);
        r->fs_kHz = c->fs_kHz; r->nb_subfr = c->nb_subfr; r->frame_length = c->frame_length; r->ltp_mem_length = c->ltp_mem_length;
        r->la_pitch = c->la_pitch; r->pitch_LPC_win_length = c->pitch_LPC_win_length; r->pitchEstimationLPCOrder = c->pitchEstimationLPCOrder;
        r->pitchEstimationComplexity = c->pitchEstimationComplexity; r->pitchEstimationThreshold_Q16 = c->pitchEstimationThreshold_Q16;
        r->signalType = c->indices.signalType; r->first_frame_after_reset = c->first_frame_after_reset; r->speech_activity_Q8 = c->speech_activity_Q8;
        r->prevSignalType = c->prevSignalType; r->input_tilt_Q15 = c->input_tilt_Q15; r->prevLag = c->prevLag; r->LTPCorr_Q15 = psEnc->LTPCorr_Q15;
    }
    __real_silk_find_pitch_lags_FIX(psEnc, psEncCtrl, res, x, arch);
    if (rec >= 0) {
        opusgpu_find_pitch_lags_out *o = &g_tout[rec];
        memcpy(o->res, res, sizeof(opus_int16) * buf_len);
        for (int k = 0; k < c->nb_subfr; k++) o->pitchL[k] = psEncCtrl->pitchL[k];
        o->lagIndex = c->indices.lagIndex; o->contourIndex = c->indices.contourIndex; o->LTPCorr_Q15 = psEnc->LTPCorr_Q15;
        o->signalType = c->indices.signalType; o->predGain_Q16 = psEncCtrl->predGain_Q16; o->status = 0;
        g_nt++;
    }
}

/* Aligned capture of one encoder run: every analysis function and both quantisers record at once. */
void refcap_start_dd(int max_records);
void refcap_start_bits(int max_records);
void refcap_start_frame(int max_records);
void refcap_start_chain(int max_records)
{
    refcap_start(max_records); refcap_start_dd(max_records); refcap_start_bits(max_records); refcap_start_fpc(max_records); refcap_start_gains(max_records);
    refcap_start_shape(max_records); refcap_start_prefilter(max_records); refcap_start_pitch(max_records); refcap_start_frame(max_records);
    g_frame = 0; g_fid_cap = max_records;
    for (int k = 0; k < FID_KINDS; k++) g_fid[k] = (int *)calloc(max_records, sizeof(int));
}

/* ---- silk_encode_indices / silk_encode_pulses (opus-fix/silk/encode_indices.c:36, encode_pulses.c:64): arguments, the fields read, and the
 * range coder (every ec_ctx field + its buffer) before and after ---- */
#include "entenc.h"
static opusgpu_silk_bits_in *g_bi_in, *g_bp_in; static opusgpu_ec_state *g_bi_ec0, *g_bi_ec1, *g_bp_ec0, *g_bp_ec1;
static opusgpu_silk_bits_out *g_bi_out, *g_bp_out; static int g_nbi, g_nbp, g_capbits;
void refcap_start_bits(int max_records)
{
    g_capbits = max_records; g_nbi = g_nbp = 0; g_on = 1;
    g_bi_in = (opusgpu_silk_bits_in *)calloc(max_records, sizeof(*g_bi_in)); g_bp_in = (opusgpu_silk_bits_in *)calloc(max_records, sizeof(*g_bp_in));
    g_bi_ec0 = (opusgpu_ec_state *)calloc(max_records, sizeof(*g_bi_ec0)); g_bi_ec1 = (opusgpu_ec_state *)calloc(max_records, sizeof(*g_bi_ec1));
    g_bp_ec0 = (opusgpu_ec_state *)calloc(max_records, sizeof(*g_bp_ec0)); g_bp_ec1 = (opusgpu_ec_state *)calloc(max_records, sizeof(*g_bp_ec1));
    g_bi_out = (opusgpu_silk_bits_out *)calloc(max_records, sizeof(*g_bi_out)); g_bp_out = (opusgpu_silk_bits_out *)calloc(max_records, sizeof(*g_bp_out));
}
int refcap_count_bits(int which) { return which == 0 ? g_nbi : g_nbp; }
int refcap_sizes_bits(int which) { return which == 0 ? sizeof(opusgpu_silk_bits_in) : which == 1 ? sizeof(opusgpu_ec_state) : sizeof(opusgpu_silk_bits_out); }
void refcap_get_bits(int which, void *in, void *ec0, void *ec1, void *out)
{
    const int n = which == 0 ? g_nbi : g_nbp;
    memcpy(in, which == 0 ? g_bi_in : g_bp_in, (size_t)n * sizeof(opusgpu_silk_bits_in));
    memcpy(ec0, which == 0 ? g_bi_ec0 : g_bp_ec0, (size_t)n * sizeof(opusgpu_ec_state));
    memcpy(ec1, which == 0 ? g_bi_ec1 : g_bp_ec1, (size_t)n * sizeof(opusgpu_ec_state));
    memcpy(out, which == 0 ? g_bi_out : g_bp_out, (size_t)n * sizeof(opusgpu_silk_bits_out));
}
static void ec_snapshot(opusgpu_ec_state *d, const ec_enc *e)
{
    memset(d, 0, sizeof(*d));
    d->storage = e->storage; d->end_offs = e->end_offs; d->end_window = e->end_window; d->nend_bits = e->nend_bits; d->nbits_total = e->nbits_total;
    d->offs = e->offs; d->rng = e->rng; d->val = e->val; d->ext = e->ext; d->rem = e->rem; d->error = e->error;
    memcpy(d->buf, e->buf, e->offs);                                                    /* head bytes */
    memcpy(d->buf + e->storage - e->end_offs, e->buf + e->storage - e->end_offs, e->end_offs);   /* tail bytes */
}

void __real_silk_encode_indices(silk_encoder_state *psEncC, ec_enc *psRangeEnc, opus_int FrameIndex, opus_int encode_LBRR, opus_int condCoding);
void __wrap_silk_encode_indices(silk_encoder_state *psEncC, ec_enc *psRangeEnc, opus_int FrameIndex, opus_int encode_LBRR, opus_int condCoding)
{
    int rec = (g_on && g_bi_in && g_nbi < g_capbits && !encode_LBRR && psRangeEnc->storage <= OPUSGPU_EC_BUF) ? g_nbi : -1;
    fid_note(FID_BITS_IDX, rec);
    if (rec >= 0) {
        opusgpu_silk_bits_in *r = &g_bi_in[rec];
        const SideInfoIndices *ix = &psEncC->indices;
        memcpy(r->GainsIndices, ix->GainsIndices, 4); memcpy(r->LTPIndex, ix->LTPIndex, 4); memcpy(r->NLSFIndices, ix->NLSFIndices, MAX_LPC_ORDER + 1);
        r->lagIndex = ix->lagIndex; r->contourIndex = ix->contourIndex; r->signalType = ix->signalType; r->quantOffsetType = ix->quantOffsetType;
        r->NLSFInterpCoef_Q2 = ix->NLSFInterpCoef_Q2; r->PERIndex = ix->PERIndex; r->LTP_scaleIndex = ix->LTP_scaleIndex; r->Seed = ix->Seed;
        r->nb_subfr = psEncC->nb_subfr; r->fs_kHz = psEncC->fs_kHz; r->predictLPCOrder = psEncC->predictLPCOrder; r->frame_length = psEncC->frame_length;
        r->condCoding = condCoding; r->ec_prevSignalType = psEncC->ec_prevSignalType; r->ec_prevLagIndex = psEncC->ec_prevLagIndex; r->which = 1;
        ec_snapshot(&g_bi_ec0[rec], psRangeEnc);
    }
    __real_silk_encode_indices(psEncC, psRangeEnc, FrameIndex, encode_LBRR, condCoding);
    if (rec >= 0) {
        ec_snapshot(&g_bi_ec1[rec], psRangeEnc);
        g_bi_out[rec].ec_prevSignalType = psEncC->ec_prevSignalType; g_bi_out[rec].ec_prevLagIndex = psEncC->ec_prevLagIndex;
        g_nbi++;
    }
}

void __real_silk_encode_pulses(ec_enc *psRangeEnc, const opus_int signalType, const opus_int quantOffsetType, opus_int8 pulses[], const opus_int frame_length);
void __wrap_silk_encode_pulses(ec_enc *psRangeEnc, const opus_int signalType, const opus_int quantOffsetType, opus_int8 pulses[], const opus_int frame_length)
{
    int rec = (g_on && g_bp_in && g_nbp < g_capbits && frame_length <= OPUSGPU_SILK_MAX_FRAME && psRangeEnc->storage <= OPUSGPU_EC_BUF) ? g_nbp : -1;
    fid_note(FID_BITS_PLS, rec);
    if (rec >= 0) {
        opusgpu_silk_bits_in *r = &g_bp_in[rec];
        memcpy(r->pulses, pulses, frame_length);
        r->signalType = signalType; r->quantOffsetType = quantOffsetType; r->frame_length = frame_length; r->which = 2;
        r->nb_subfr = 4; r->fs_kHz = 16; r->predictLPCOrder = 16;                       /* not read by silk_encode_pulses; valid placeholders */
        ec_snapshot(&g_bp_ec0[rec], psRangeEnc);
    }
    __real_silk_encode_pulses(psRangeEnc, signalType, quantOffsetType, pulses, frame_length);
    if (rec >= 0) { ec_snapshot(&g_bp_ec1[rec], psRangeEnc); g_nbp++; }
}

/* ---- silk_VAD_GetSA_Q8_c (opus-fix/silk/VAD.c:82): the input frame and silk_VAD_state before / after -> speech activity, tilt, input quality ---- */
static opusgpu_vad_in *g_vin; static opusgpu_vad_state *g_vst0, *g_vst1; static opusgpu_vad_out *g_vout; static int g_nv, g_capv;
void refcap_start_vad(int max_records)
{
    g_capv = max_records; g_nv = 0; g_on = 1;
    g_vin = (opusgpu_vad_in *)calloc(max_records, sizeof(*g_vin)); g_vst0 = (opusgpu_vad_state *)calloc(max_records, sizeof(*g_vst0));
    g_vst1 = (opusgpu_vad_state *)calloc(max_records, sizeof(*g_vst1)); g_vout = (opusgpu_vad_out *)calloc(max_records, sizeof(*g_vout));
}
int refcap_count_vad(void) { return g_nv; }
int refcap_sizes_vad(int which) { return which == 0 ? sizeof(opusgpu_vad_in) : which == 1 ? sizeof(opusgpu_vad_state) : which == 2 ? sizeof(opusgpu_vad_out) : (int)sizeof(silk_VAD_state); }
void refcap_get_vad(void *vin, void *st0, void *st1, void *vout)
{
    memcpy(vin, g_vin, (size_t)g_nv * sizeof(*g_vin)); memcpy(st0, g_vst0, (size_t)g_nv * sizeof(*g_vst0));
    memcpy(st1, g_vst1, (size_t)g_nv * sizeof(*g_vst1)); memcpy(vout, g_vout, (size_t)g_nv * sizeof(*g_vout));
}
opus_int __real_silk_VAD_GetSA_Q8_c(silk_encoder_state *psEncC, const opus_int16 pIn[]);
opus_int __wrap_silk_VAD_GetSA_Q8_c(silk_encoder_state *psEncC, const opus_int16 pIn[])
{
    typedef char vad_state_layout[sizeof(opusgpu_vad_state) == sizeof(silk_VAD_state) ? 1 : -1];
    int rec = (g_on && g_vin && g_nv < g_capv && psEncC->frame_length <= OPUSGPU_SILK_MAX_FRAME) ? g_nv : -1;
    if (rec >= 0) {
        memcpy(g_vin[rec].pIn, pIn, sizeof(opus_int16) * psEncC->frame_length);
        g_vin[rec].frame_length = psEncC->frame_length; g_vin[rec].fs_kHz = psEncC->fs_kHz;
        memcpy(&g_vst0[rec], &psEncC->sVAD, sizeof(g_vst0[rec]));
    }
    const opus_int ret = __real_silk_VAD_GetSA_Q8_c(psEncC, pIn);
    if (rec >= 0) {
        memcpy(&g_vst1[rec], &psEncC->sVAD, sizeof(g_vst1[rec]));
        g_vout[rec].speech_activity_Q8 = psEncC->speech_activity_Q8; g_vout[rec].input_tilt_Q15 = psEncC->input_tilt_Q15;
        for (int k = 0; k < VAD_N_BANDS; k++) g_vout[rec].input_quality_bands_Q15[k] = psEncC->input_quality_bands_Q15[k];
        g_nv++;
    }
    return ret;
}

/* ---- silk_encode_frame_FIX whole (opus-fix/silk/fixed/encode_frame_FIX.c:88): its arguments and what the bitrate loop leaves behind --
 * the range coder, the quantiser state, the gain index state, the pulses -- plus the number of quantiser passes it took. Part of the
 * aligned capture (frame id = the find_pitch_lags call it contains). Records of this file only (test-side layout):
 *   args[4]  = condCoding, maxBits, useCBR, quantiser passes
 *   misc     = pulses[320], GainsIndices[4], LastGainIndex, Seed, nBytesOut, prefill, reserved[4] */
typedef struct { opus_int8 pulses[OPUSGPU_SILK_MAX_FRAME]; opus_int8 GainsIndices[4]; opus_int32 LastGainIndex, Seed, nBytesOut, prefill, reserved[4]; } refcap_frame_misc;
static opus_int32 (*g_fr_args)[4]; static opusgpu_ec_state *g_fr_ec; static opusgpu_nsq_state *g_fr_nsq; static refcap_frame_misc *g_fr_misc;
static int g_nfr, g_capfr;
void refcap_start_frame(int max_records)
{
    g_capfr = max_records; g_nfr = 0; g_on = 1;
    g_fr_args = calloc(max_records, sizeof(*g_fr_args)); g_fr_ec = calloc(max_records, sizeof(*g_fr_ec));
    g_fr_nsq = calloc(max_records, sizeof(*g_fr_nsq)); g_fr_misc = calloc(max_records, sizeof(*g_fr_misc));
}
int refcap_count_frame(void) { return g_nfr; }
int refcap_sizes_frame(int which) { return which == 0 ? (int)sizeof(*g_fr_args) : which == 1 ? (int)sizeof(*g_fr_ec) : which == 2 ? (int)sizeof(*g_fr_nsq) : (int)sizeof(*g_fr_misc); }
void refcap_get_frame(void *args, void *ec, void *nsq, void *misc)
{
    memcpy(args, g_fr_args, (size_t)g_nfr * sizeof(*g_fr_args)); memcpy(ec, g_fr_ec, (size_t)g_nfr * sizeof(*g_fr_ec));
    memcpy(nsq, g_fr_nsq, (size_t)g_nfr * sizeof(*g_fr_nsq)); memcpy(misc, g_fr_misc, (size_t)g_nfr * sizeof(*g_fr_misc));
}
opus_int __real_silk_encode_frame_FIX(silk_encoder_state_FIX *psEnc, opus_int32 *pnBytesOut, ec_enc *psRangeEnc, opus_int condCoding, opus_int maxBits, opus_int useCBR);
opus_int __wrap_silk_encode_frame_FIX(silk_encoder_state_FIX *psEnc, opus_int32 *pnBytesOut, ec_enc *psRangeEnc, opus_int condCoding, opus_int maxBits, opus_int useCBR)
{
    const silk_encoder_state *c = &psEnc->sCmn;
    const int prefill = c->prefillFlag;
    int rec = (g_on && g_fr_args && g_nfr < g_capfr && !prefill && psRangeEnc->storage <= OPUSGPU_EC_BUF && c->frame_length <= OPUSGPU_SILK_MAX_FRAME) ? g_nfr : -1;
    const int q0 = g_q_calls;
    if (rec >= 0 && g_fid[FID_FRAME] && rec < g_fid_cap) g_fid[FID_FRAME][rec] = g_frame + 1;      /* the pitch analysis inside counts the frame */
    opus_int ret = __real_silk_encode_frame_FIX(psEnc, pnBytesOut, psRangeEnc, condCoding, maxBits, useCBR);
    if (rec >= 0) {
        g_fr_args[rec][0] = condCoding; g_fr_args[rec][1] = maxBits; g_fr_args[rec][2] = useCBR; g_fr_args[rec][3] = g_q_calls - q0;
        ec_snapshot(&g_fr_ec[rec], psRangeEnc);
        memcpy(&g_fr_nsq[rec], &c->sNSQ, sizeof(g_fr_nsq[rec]));
        refcap_frame_misc *m = &g_fr_misc[rec];
        memcpy(m->pulses, c->pulses, c->frame_length); memcpy(m->GainsIndices, c->indices.GainsIndices, 4);
        m->LastGainIndex = psEnc->sShape.LastGainIndex; m->Seed = c->indices.Seed; m->nBytesOut = *pnBytesOut; m->prefill = prefill;
        g_nfr++;
    }
    return ret;
}

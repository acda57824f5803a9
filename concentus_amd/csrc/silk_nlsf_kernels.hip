// silk_nlsf_kernels.hip -- batched silk_process_NLSFs (opus-fix/silk/process_NLSFs.c:35-106) and silk_residual_energy_FIX
// (opus-fix/silk/fixed/residual_energy_FIX.c:37-98): the two calls silk_find_pred_coefs_FIX makes after silk_find_LPC_FIX
// (find_pred_coefs_FIX.c:139-143). One lane per record; the arithmetic lives in silk_nlsf_dev.h.
#include <string.h>
#include "silk_nlsf_dev.h"
#include "opusgpu_internal.h"
#include "../../include/opusgpu_silk.h"
#include "../../include/opusgpu_hooks.h"
#include "silk_validate.h"

namespace ca {

__global__ __launch_bounds__(64) void silk_process_nlsfs_kernel(const opusgpu_process_nlsf_in *__restrict__ recs,
                                                                opusgpu_process_nlsf_out *__restrict__ outs, int n_rec,
                                                                int *__restrict__ bad_records)
{
    __shared__ NlsfTablesLds tables;
    __shared__ NlsfEncTables enc;
    nlsf_stage_tables(tables, threadIdx.x, blockDim.x);
    nlsf_stage_enc_tables(enc, threadIdx.x, blockDim.x);
    __syncthreads();
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rec) return;
    const opusgpu_process_nlsf_in in = recs[r];
    opusgpu_process_nlsf_out o;
    memset(&o, 0, sizeof(o));
    if (!process_nlsf_record_ok(in)) {
        o.status = OPUSGPU_BAD_ARG;
        outs[r] = o;
        atomicAdd(bad_records, 1);
        return;
    }
    i16 nlsf[SILK_MAX_LPC], prev[SILK_MAX_LPC], pc[2][SILK_MAX_LPC];
    i8 idx[SILK_MAX_LPC + 1];
    for (int k = 0; k < SILK_MAX_LPC; k++) { nlsf[k] = in.NLSF_Q15[k]; prev[k] = in.prev_NLSFq_Q15[k]; pc[0][k] = pc[1][k] = 0; }
    for (int k = 0; k <= SILK_MAX_LPC; k++) idx[k] = 0;
    silk_process_NLSFs_dev(pc, idx, nlsf, prev, in.speech_activity_Q8, in.nb_subfr, in.predictLPCOrder, in.useInterpolatedNLSFs,
                           in.NLSFInterpCoef_Q2, in.NLSF_MSVQ_Survivors, in.signalType, &tables, &enc);
    for (int k = 0; k < in.predictLPCOrder; k++) { o.PredCoef_Q12[0][k] = pc[0][k]; o.PredCoef_Q12[1][k] = pc[1][k]; o.NLSF_Q15[k] = nlsf[k]; }
    for (int k = 0; k <= in.predictLPCOrder; k++) o.NLSFIndices[k] = idx[k];
    o.status = OPUSGPU_OK;
    outs[r] = o;
}

struct ResX {                                                  // this lane's column of the [sample][lane] block
    const i16 *p;
    __device__ __forceinline__ i32 operator[](int k) const { return p[k * 64]; }
};

__global__ __launch_bounds__(64) void silk_residual_energy_kernel(const opusgpu_res_nrg_in *__restrict__ recs,
                                                                  opusgpu_res_nrg_out *__restrict__ outs, int n_rec,
                                                                  int *__restrict__ bad_records)
{
    i16 xs[OPUSGPU_SILK_BURG_MAX_X];                           // private: the residual filter's register window reads every sample once
    const int r = blockIdx.x * 64 + threadIdx.x;
    if (r >= n_rec) return;
    const opusgpu_res_nrg_in &in = recs[r];
    opusgpu_res_nrg_out o;
    memset(&o, 0, sizeof(o));
    if (!res_nrg_record_ok(in)) {
        o.status = OPUSGPU_BAD_ARG;
        outs[r] = o;
        atomicAdd(bad_records, 1);
        return;
    }
    {
        static_assert(sizeof(opusgpu_res_nrg_in) % 16 == 0, "16-byte loads of x");
        const int nx = (in.subfr_length + in.LPC_order) * in.nb_subfr;
        const int4 *src = reinterpret_cast<const int4 *>(in.x);
        for (int k = 0; k < nx; k += 8) {
            const int4 w = src[k >> 3];
            xs[k + 0] = (i16)w.x; xs[k + 1] = (i16)(w.x >> 16); xs[k + 2] = (i16)w.y; xs[k + 3] = (i16)(w.y >> 16);
            xs[k + 4] = (i16)w.z; xs[k + 5] = (i16)(w.z >> 16); xs[k + 6] = (i16)w.w; xs[k + 7] = (i16)(w.w >> 16);
        }
    }
    const i16 *x = xs;
    i16 a[2][SILK_MAX_LPC];
    i32 gains[4], nrgs[4] = {0, 0, 0, 0}, nrgsQ[4] = {0, 0, 0, 0};
    for (int k = 0; k < SILK_MAX_LPC; k++) { a[0][k] = in.a_Q12[0][k]; a[1][k] = in.a_Q12[1][k]; }
    for (int k = 0; k < 4; k++) gains[k] = in.gains[k];
    silk_residual_energy_dev(nrgs, nrgsQ, x, a, gains, in.subfr_length, in.nb_subfr, in.LPC_order);
    for (int k = 0; k < in.nb_subfr; k++) { o.nrgs[k] = nrgs[k]; o.nrgsQ[k] = nrgsQ[k]; }
    o.status = OPUSGPU_OK;
    outs[r] = o;
}

}  // namespace ca

using namespace ca;

extern "C" int opusgpu_silk_process_nlsfs_batch(const opusgpu_process_nlsf_in *d_in, opusgpu_process_nlsf_out *d_out, int n, void *stream)
{
    if (n < 0) return OPUSGPU_BAD_ARG;
    if (n == 0) return OPUSGPU_OK;
    if (!d_in || !d_out) return OPUSGPU_BAD_ARG;
    int *bad = opusgpu_bad_record_counter();
    if (!bad) return OPUSGPU_ALLOC_FAIL;
    const int lpb = opusgpu_silk_lanes_per_block();
    hipLaunchKernelGGL(silk_process_nlsfs_kernel, dim3((n + lpb - 1) / lpb), dim3(lpb), 0, (hipStream_t)stream, d_in, d_out, n, bad);
    return opusgpu_check_launch();
}

extern "C" int opusgpu_silk_residual_energy_batch(const opusgpu_res_nrg_in *d_in, opusgpu_res_nrg_out *d_out, int n, void *stream)
{
    if (n < 0) return OPUSGPU_BAD_ARG;
    if (n == 0) return OPUSGPU_OK;
    if (!d_in || !d_out) return OPUSGPU_BAD_ARG;
    int *bad = opusgpu_bad_record_counter();
    if (!bad) return OPUSGPU_ALLOC_FAIL;
    hipLaunchKernelGGL(silk_residual_energy_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, d_in, d_out, n, bad);
    return opusgpu_check_launch();
}

// ---- per-call hooks with the reference's own argument lists ------------------------------------------------------------
static int rd_int(const void *base, int off) { int v; memcpy(&v, (const char *)base + off, sizeof(v)); return v; }

template <class In, class Out, class Launch>
static int run_one(const In &h_in, Out &h_out, Launch launch)
{
    In *d_in = nullptr;
    Out *d_out = nullptr;
    if (hipMalloc(&d_in, sizeof(In)) != hipSuccess || hipMalloc(&d_out, sizeof(Out)) != hipSuccess) {
        if (d_in) (void)hipFree(d_in);
        return OPUSGPU_ALLOC_FAIL;
    }
    int rc = hipMemcpy(d_in, &h_in, sizeof(In), hipMemcpyHostToDevice) == hipSuccess ? OPUSGPU_OK : OPUSGPU_INTERNAL_ERROR;
    OpusgpuHookBadScope bad;                 // rejected records count into this thread's counter, not the device's shared one
    if (rc == OPUSGPU_OK) rc = bad.rc;
    if (rc == OPUSGPU_OK) rc = launch(d_in, d_out);
    if (rc == OPUSGPU_OK && hipMemcpy(&h_out, d_out, sizeof(Out), hipMemcpyDeviceToHost) != hipSuccess) rc = OPUSGPU_INTERNAL_ERROR;
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    if (rc == OPUSGPU_OK && h_out.status != OPUSGPU_OK) rc = h_out.status;
    return rc;
}

// silk_process_NLSFs(psEncC, PredCoef_Q12, pNLSF_Q15, prev_NLSFq_Q15) (silk/main.h; called at silk/fixed/find_pred_coefs_FIX.c:139)
extern "C" void opusgpu_silk_process_NLSFs(void *psEncC, int16_t PredCoef_Q12[/*2 * 16*/], int16_t pNLSF_Q15[], const int16_t prev_NLSFq_Q15[])
{
    if (!psEncC || !PredCoef_Q12 || !pNLSF_Q15 || !prev_NLSFq_Q15) { opusgpu_set_last_error(OPUSGPU_BAD_ARG); return; }
    opusgpu_process_nlsf_in h_in;
    opusgpu_process_nlsf_out h_out;
    memset(&h_in, 0, sizeof(h_in));
    h_in.speech_activity_Q8 = rd_int(psEncC, OPUSGPU_REF_OFF_SPEECH_ACTIVITY_Q8);
    h_in.nb_subfr = rd_int(psEncC, OPUSGPU_REF_OFF_NB_SUBFR);
    h_in.predictLPCOrder = rd_int(psEncC, OPUSGPU_REF_OFF_PREDICT_LPC_ORDER);
    h_in.useInterpolatedNLSFs = rd_int(psEncC, OPUSGPU_REF_OFF_USE_INTERPOLATED_NLSFS);
    h_in.NLSF_MSVQ_Survivors = rd_int(psEncC, OPUSGPU_REF_OFF_NLSF_MSVQ_SURVIVORS);
    const int8_t *indices = (const int8_t *)psEncC + OPUSGPU_REF_OFF_INDICES;
    h_in.NLSFInterpCoef_Q2 = indices[OPUSGPU_REF_OFF_NLSF_INTERP_COEF_Q2];
    h_in.signalType = indices[OPUSGPU_REF_OFF_SIGNAL_TYPE];
    const int D = h_in.predictLPCOrder;
    if (D != 10 && D != 16) { opusgpu_set_last_error(OPUSGPU_BAD_ARG); return; }
    memcpy(h_in.NLSF_Q15, pNLSF_Q15, sizeof(int16_t) * (size_t)D);
    memcpy(h_in.prev_NLSFq_Q15, prev_NLSFq_Q15, sizeof(int16_t) * (size_t)D);
    const int rc = run_one(h_in, h_out, [](const opusgpu_process_nlsf_in *i, opusgpu_process_nlsf_out *o) {
        return opusgpu_silk_process_nlsfs_batch(i, o, 1, nullptr);
    });
    opusgpu_set_last_error(rc);
    if (rc != OPUSGPU_OK) return;
    memcpy(PredCoef_Q12, h_out.PredCoef_Q12[0], sizeof(int16_t) * (size_t)D);
    memcpy(PredCoef_Q12 + OPUSGPU_SILK_MAX_ORDER, h_out.PredCoef_Q12[1], sizeof(int16_t) * (size_t)D);
    memcpy(pNLSF_Q15, h_out.NLSF_Q15, sizeof(int16_t) * (size_t)D);
    memcpy((int8_t *)psEncC + OPUSGPU_REF_OFF_INDICES + OPUSGPU_REF_OFF_NLSF_INDICES, h_out.NLSFIndices, (size_t)D + 1);
}

// silk_residual_energy_FIX(nrgs, nrgsQ, x, a_Q12, gains, subfr_length, nb_subfr, LPC_order, arch) (silk/fixed/main_FIX.h; called at
// silk/fixed/find_pred_coefs_FIX.c:142)
extern "C" void opusgpu_silk_residual_energy_FIX(int32_t nrgs[], int nrgsQ[], const int16_t x[], int16_t a_Q12[/*2 * 16*/], const int32_t gains[],
                                                 const int subfr_length, const int nb_subfr, const int LPC_order, int arch)
{
    (void)arch;
    if (!nrgs || !nrgsQ || !x || !a_Q12 || !gains) { opusgpu_set_last_error(OPUSGPU_BAD_ARG); return; }
    const long nx = (long)(subfr_length + LPC_order) * nb_subfr;
    if ((nb_subfr != 2 && nb_subfr != 4) || subfr_length < 1 || LPC_order < 2 || LPC_order > 16 || nx > OPUSGPU_SILK_BURG_MAX_X) {
        opusgpu_set_last_error(OPUSGPU_BAD_ARG);
        return;
    }
    opusgpu_res_nrg_in h_in;
    opusgpu_res_nrg_out h_out;
    memset(&h_in, 0, sizeof(h_in));
    memcpy(h_in.x, x, sizeof(int16_t) * (size_t)nx);
    memcpy(h_in.a_Q12[0], a_Q12, sizeof(int16_t) * (size_t)LPC_order);
    memcpy(h_in.a_Q12[1], a_Q12 + OPUSGPU_SILK_MAX_ORDER, sizeof(int16_t) * (size_t)LPC_order);
    memcpy(h_in.gains, gains, sizeof(int32_t) * (size_t)nb_subfr);
    h_in.subfr_length = subfr_length; h_in.nb_subfr = nb_subfr; h_in.LPC_order = LPC_order;
    const int rc = run_one(h_in, h_out, [](const opusgpu_res_nrg_in *i, opusgpu_res_nrg_out *o) {
        return opusgpu_silk_residual_energy_batch(i, o, 1, nullptr);
    });
    opusgpu_set_last_error(rc);
    if (rc != OPUSGPU_OK) return;
    for (int k = 0; k < nb_subfr; k++) { nrgs[k] = h_out.nrgs[k]; nrgsQ[k] = h_out.nrgsQ[k]; }
}

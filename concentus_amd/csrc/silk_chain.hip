// silk_chain.hip -- the first pass of silk_encode_frame_FIX (opus-fix/silk/fixed/encode_frame_FIX.c:176-336) for a batch of frames as
// ONE C entry point: the eight batched kernels back to back on one stream, and between them small kernels that complete the next
// records from the outputs of the earlier stages -- the assignments the reference makes through psEnc / psEncCtrl, here field to
// field between the flat records of include/opusgpu_silk.h (the same edges concentus_amd/silk_chain.py documents and
// tests/test_silk_chain_cpu.py pins against the reference). No arithmetic happens in this file.
#include <hip/hip_runtime.h>
#include "opusgpu_internal.h"
#include "../../include/opusgpu_silk.h"

namespace {

// one 64-thread workgroup per frame: arrays copied cooperatively, scalars by thread 0
template <class T> __device__ __forceinline__ void cp(T *dst, const T *src, int n) { for (int k = threadIdx.x; k < n; k += 64) dst[k] = src[k]; }
__device__ __forceinline__ opusgpu_nsq_in &qrec(void *q, size_t stride, int r) { return *(opusgpu_nsq_in *)((char *)q + stride * (size_t)r); }

__global__ __launch_bounds__(64) void after_pitch_kernel(opusgpu_silk_chain_bufs b, size_t qs, int fl, int ltp, int n)
{
    const int r = blockIdx.x;
    if (r >= n) return;
    const opusgpu_find_pitch_lags_out &p = b.pitch_out[r];
    cp(b.shape_in[r].pitch_res, p.res + ltp, fl);
    cp(b.fpc_in[r].res_pitch, p.res, ltp + fl);
    opusgpu_nsq_in &q = qrec(b.q_in, qs, r);
    cp(b.shape_in[r].pitchL, p.pitchL, 4); cp(b.fpc_in[r].pitchL, p.pitchL, 4); cp(b.prefilter_in[r].pitchL, p.pitchL, 4); cp(q.pitchL, p.pitchL, 4);
    if (threadIdx.x == 0) {
        b.shape_in[r].signalType = p.signalType; b.shape_in[r].LTPCorr_Q15 = p.LTPCorr_Q15; b.shape_in[r].predGain_Q16 = p.predGain_Q16;
        b.fpc_in[r].signalType = p.signalType; b.gains_in[r].signalType = p.signalType; b.prefilter_in[r].signalType = p.signalType;
        q.signalType = p.signalType;
        if (b.bits_in) { b.bits_in[r].lagIndex = p.lagIndex; b.bits_in[r].contourIndex = p.contourIndex; b.bits_in[r].signalType = p.signalType; }
    }
}

__global__ __launch_bounds__(64) void after_shape_kernel(opusgpu_silk_chain_bufs b, size_t qs, int n)
{
    const int r = blockIdx.x;
    if (r >= n) return;
    const opusgpu_noise_shape_out &s = b.shape_out[r];
    opusgpu_nsq_in &q = qrec(b.q_in, qs, r);
    opusgpu_prefilter_in &x = b.prefilter_in[r];
    cp(x.AR1_Q13, s.AR1_Q13, 64); cp(q.AR2_Q13, s.AR2_Q13, 64);
    cp(b.fpc_in[r].Gains_Q16, s.Gains_Q16, 4); cp(b.gains_in[r].Gains_Q16, s.Gains_Q16, 4);
    cp(x.HarmShapeGain_Q14, s.HarmShapeGain_Q14, 4); cp(x.HarmBoost_Q14, s.HarmBoost_Q14, 4); cp(x.Tilt_Q14, s.Tilt_Q14, 4);
    cp(x.GainsPre_Q14, s.GainsPre_Q14, 4); cp(x.LF_shp_Q14, s.LF_shp_Q14, 4);
    cp(q.HarmShapeGain_Q14, s.HarmShapeGain_Q14, 4); cp(q.Tilt_Q14, s.Tilt_Q14, 4); cp(q.LF_shp_Q14, s.LF_shp_Q14, 4);
    if (threadIdx.x == 0) {
        b.fpc_in[r].coding_quality_Q14 = s.coding_quality_Q14; x.coding_quality_Q14 = s.coding_quality_Q14;
        b.gains_in[r].quantOffsetType = s.quantOffsetType; b.gains_in[r].input_quality_Q14 = s.input_quality_Q14;
        b.gains_in[r].coding_quality_Q14 = s.coding_quality_Q14;
    }
}

__global__ __launch_bounds__(64) void after_pred_coefs_kernel(opusgpu_silk_chain_bufs b, size_t qs, int n)
{
    const int r = blockIdx.x;
    if (r >= n) return;
    const opusgpu_find_pred_coefs_out &f = b.fpc_out[r];
    opusgpu_nsq_in &q = qrec(b.q_in, qs, r);
    cp(q.PredCoef_Q12, &f.PredCoef_Q12[0][0], 32); cp(q.LTPCoef_Q14, f.LTPCoef_Q14, 20);
    cp(b.gains_in[r].ResNrg, f.ResNrg, 4); cp(b.gains_in[r].ResNrgQ, f.ResNrgQ, 4);
    if (b.bits_in) { cp(b.bits_in[r].NLSFIndices, f.NLSFIndices, OPUSGPU_SILK_MAX_ORDER + 1); cp(b.bits_in[r].LTPIndex, f.LTPIndex, 4); }
    if (threadIdx.x == 0) {
        b.gains_in[r].LTPredCodGain_Q7 = f.LTPredCodGain_Q7;
        q.LTP_scale_Q14 = f.LTP_scale_Q14; q.NLSFInterpCoef_Q2 = f.NLSFInterpCoef_Q2;
        if (b.bits_in) { b.bits_in[r].NLSFInterpCoef_Q2 = f.NLSFInterpCoef_Q2; b.bits_in[r].PERIndex = f.PERIndex; b.bits_in[r].LTP_scaleIndex = f.LTP_scaleIndex; }
    }
}

__global__ __launch_bounds__(64) void after_gains_kernel(opusgpu_silk_chain_bufs b, size_t qs, int n)
{
    const int r = blockIdx.x;
    if (r >= n) return;
    const opusgpu_process_gains_out &g = b.gains_out[r];
    opusgpu_nsq_in &q = qrec(b.q_in, qs, r);
    cp(q.Gains_Q16, g.Gains_Q16, 4);
    if (b.bits_in) cp(b.bits_in[r].GainsIndices, g.GainsIndices, 4);
    if (threadIdx.x == 0) {
        q.Lambda_Q10 = g.Lambda_Q10; q.quantOffsetType = g.quantOffsetType;
        if (b.bits_in) b.bits_in[r].quantOffsetType = g.quantOffsetType;
    }
}

__global__ __launch_bounds__(64) void after_prefilter_kernel(opusgpu_silk_chain_bufs b, size_t qs, int n)
{
    const int r = blockIdx.x;
    if (r >= n) return;
    cp(qrec(b.q_in, qs, r).x_Q3, b.prefilter_out[r].xw_Q3, OPUSGPU_SILK_MAX_FRAME);
}

__global__ __launch_bounds__(64) void after_quantiser_kernel(opusgpu_silk_chain_bufs b, int del_dec, int n)
{
    const int r = blockIdx.x;
    if (r >= n) return;
    static_assert(sizeof(opusgpu_nsq_out) == OPUSGPU_SILK_MAX_FRAME && offsetof(opusgpu_nsq_dd_out, pulses) == 0, "pulses first");
    const int8_t *pulses = del_dec ? ((const opusgpu_nsq_dd_out *)b.q_out)[r].pulses : ((const opusgpu_nsq_out *)b.q_out)[r].pulses;
    cp((int32_t *)b.bits_in[r].pulses, (const int32_t *)pulses, OPUSGPU_SILK_MAX_FRAME / 4);
    if (del_dec && threadIdx.x == 0) b.bits_in[r].Seed = ((const opusgpu_nsq_dd_out *)b.q_out)[r].Seed;   // NSQ_del_dec.c:297
}

}  // namespace

extern "C" int opusgpu_silk_encode_frames_batch(const opusgpu_silk_chain_bufs *bufs, int fs_kHz, int nb_subfr, int del_dec, int n, void *stream)
{
    if (!bufs || n < 0) return OPUSGPU_BAD_ARG;
    if (n == 0) return OPUSGPU_OK;
    const opusgpu_silk_chain_bufs b = *bufs;
    if (!b.pitch_in || !b.pitch_out || !b.shape_in || !b.shape_out || !b.fpc_in || !b.fpc_out || !b.gains_in || !b.gains_out || !b.prefilter_in ||
        !b.prefilter_state || !b.prefilter_out || !b.q_in || !b.nsq_state || !b.q_out || (b.bits_in && (!b.ec_state || !b.bits_out)))
        return OPUSGPU_BAD_ARG;
    if (!((fs_kHz == 8 || fs_kHz == 16) && (nb_subfr == 2 || nb_subfr == 4))) return OPUSGPU_BAD_ARG;
    const int fl = 5 * fs_kHz * nb_subfr, ltp = 20 * fs_kHz;
    const size_t qs = del_dec ? sizeof(opusgpu_nsq_dd_in) : sizeof(opusgpu_nsq_in);
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(n), block(64);
    int rc;
#define CA_STEP(call) do { rc = (call); if (rc != OPUSGPU_OK) return rc; } while (0)
#define CA_MOVE(kernel, ...) do { hipLaunchKernelGGL(kernel, grid, block, 0, s, __VA_ARGS__); rc = opusgpu_check_launch(); if (rc != OPUSGPU_OK) return rc; } while (0)
    CA_STEP(opusgpu_silk_find_pitch_lags_batch(b.pitch_in, b.pitch_out, n, stream));
    CA_MOVE(after_pitch_kernel, b, qs, fl, ltp, n);
    CA_STEP(opusgpu_silk_noise_shape_analysis_batch(b.shape_in, b.shape_out, n, stream));
    CA_MOVE(after_shape_kernel, b, qs, n);
    CA_STEP(opusgpu_silk_find_pred_coefs_batch(b.fpc_in, b.fpc_out, n, stream));
    CA_MOVE(after_pred_coefs_kernel, b, qs, n);
    CA_STEP(opusgpu_silk_process_gains_batch(b.gains_in, b.gains_out, n, stream));
    CA_MOVE(after_gains_kernel, b, qs, n);
    CA_STEP(opusgpu_silk_prefilter_batch(b.prefilter_in, b.prefilter_state, b.prefilter_out, n, stream));
    CA_MOVE(after_prefilter_kernel, b, qs, n);
    if (del_dec)
        CA_STEP(opusgpu_silk_nsq_del_dec_batch((const opusgpu_nsq_dd_in *)b.q_in, b.nsq_state, (opusgpu_nsq_dd_out *)b.q_out, n, b.workspace, b.workspace_bytes, stream));
    else
        CA_STEP(opusgpu_silk_nsq_batch((const opusgpu_nsq_in *)b.q_in, b.nsq_state, (opusgpu_nsq_out *)b.q_out, n, b.workspace, b.workspace_bytes, stream));
    if (b.bits_in) {
        CA_MOVE(after_quantiser_kernel, b, del_dec, n);
        CA_STEP(opusgpu_silk_encode_bits_batch(b.bits_in, b.ec_state, b.bits_out, n, stream));
    }
#undef CA_STEP
#undef CA_MOVE
    return OPUSGPU_OK;
}

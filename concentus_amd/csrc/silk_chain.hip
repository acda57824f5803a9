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

// ---- the bitrate loop of silk_encode_frame_FIX (encode_frame_FIX.c:263-423) on the device ---------------------------------------
// Per-frame working copies (the reference's sNSQ_copy / sRangeEnc_copy = the state a frame entered with, sNSQ_copy2 /
// sRangeEnc_copy2 + ec_buf_copy = the kept lower bracket) and the list of frames that take another pass.
struct LoopWs {
    opusgpu_nsq_state *nsq_entry, *nsq_low;
    opusgpu_ec_state *ec_entry, *ec_low;
    int *rows;                    // frames asking for another pass, in no particular order (frames are independent)
    int *count;                   // [0] = length of rows, [1] = frames with status != OK
};

template <class T> __device__ __forceinline__ void cp16(T *dst, const T *src)          // a whole record, word by word (4 380 / 1 328 bytes)
{
    static_assert(sizeof(T) % 4 == 0, "records move in 4-byte units");
    const int32_t *s = reinterpret_cast<const int32_t *>(src);
    int32_t *d = reinterpret_cast<int32_t *>(dst);
    for (int k = threadIdx.x; k < (int)(sizeof(T) / 4); k += 64) d[k] = s[k];
}

// before the first pass: keep what every frame enters with (:272-273)
__global__ __launch_bounds__(64) void loop_entry_kernel(opusgpu_silk_chain_bufs b, LoopWs w, int n)
{
    const int r = blockIdx.x;
    if (r >= n) return;
    cp16(&w.nsq_entry[r], &b.nsq_state[r]);
    cp16(&w.ec_entry[r], &b.ec_state[r]);
}

// after the first pass: what silk_process_gains_FIX left is where the loop starts (:263-270)
__global__ __launch_bounds__(64) void loop_init_kernel(opusgpu_silk_chain_bufs b, opusgpu_silk_rate_ctl *ctl, int n)
{
    const int r = blockIdx.x * 64 + threadIdx.x;
    if (r >= n) return;
    const opusgpu_process_gains_out &g = b.gains_out[r];
    opusgpu_silk_rate_ctl &c = ctl[r];
    for (int k = 0; k < 4; k++) { c.GainsUnq_Q16[k] = g.GainsUnq_Q16[k]; c.Gains_Q16[k] = g.Gains_Q16[k]; c.GainsIndices[k] = g.GainsIndices[k]; }
    c.lastGainIndexPrev = g.lastGainIndexPrev; c.LastGainIndex = g.LastGainIndex; c.Lambda_Q10 = g.Lambda_Q10;
}

// after a rate-control step: act on its decisions. One workgroup per frame; a frame that is done and neither saves nor restores
// costs one read. save2: keep the pass just consumed as the lower bracket (:389-395); restore2: the kept bracket is the result
// (:361-368); recode: back to the entry state (:283-289), next pass's gains / Lambda into the quantiser's record, onto the list.
__global__ __launch_bounds__(64) void loop_act_kernel(opusgpu_silk_chain_bufs b, const opusgpu_silk_rate_ctl *ctl, LoopWs w, size_t qs, int n)
{
    const int r = blockIdx.x;
    if (r >= n) return;
    const opusgpu_silk_rate_ctl &c = ctl[r];
    if (c.save2) { cp16(&w.nsq_low[r], &b.nsq_state[r]); cp16(&w.ec_low[r], &b.ec_state[r]); }
    if (c.restore2) { cp16(&b.nsq_state[r], &w.nsq_low[r]); cp16(&b.ec_state[r], &w.ec_low[r]); }
    if (c.recode) {
        cp16(&b.nsq_state[r], &w.nsq_entry[r]);
        cp16(&b.ec_state[r], &w.ec_entry[r]);
        opusgpu_nsq_in &q = qrec(b.q_in, qs, r);
        cp(q.Gains_Q16, c.Gains_Q16, 4);
        cp(b.bits_in[r].GainsIndices, c.GainsIndices, 4);
        if (threadIdx.x == 0) {
            q.Lambda_Q10 = c.Lambda_Q10;
            w.rows[atomicAdd(&w.count[0], 1)] = r;
        }
    }
    if (threadIdx.x == 0 && c.status != OPUSGPU_OK) atomicAdd(&w.count[1], 1);
}

// after_quantiser_kernel for a list of frames
__global__ __launch_bounds__(64) void after_quantiser_rows_kernel(opusgpu_silk_chain_bufs b, int del_dec, const int *rows, int m)
{
    if ((int)blockIdx.x >= m) return;
    const int r = rows[blockIdx.x];
    const int8_t *pulses = del_dec ? ((const opusgpu_nsq_dd_out *)b.q_out)[r].pulses : ((const opusgpu_nsq_out *)b.q_out)[r].pulses;
    cp((int32_t *)b.bits_in[r].pulses, (const int32_t *)pulses, OPUSGPU_SILK_MAX_FRAME / 4);
    if (del_dec && threadIdx.x == 0) b.bits_in[r].Seed = ((const opusgpu_nsq_dd_out *)b.q_out)[r].Seed;
}

}  // namespace

extern "C" int opusgpu_silk_nsq_rows(const opusgpu_nsq_in *, opusgpu_nsq_state *, opusgpu_nsq_out *, const int *, int, void *, hipStream_t);
extern "C" int opusgpu_silk_nsq_del_dec_rows(const opusgpu_nsq_dd_in *, opusgpu_nsq_state *, opusgpu_nsq_dd_out *, const int *, int, void *, hipStream_t);
extern "C" int opusgpu_silk_encode_bits_rows(const opusgpu_silk_bits_in *, opusgpu_ec_state *, opusgpu_silk_bits_out *, const int *, int, hipStream_t);

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

// ---- silk_encode_frame_FIX with its bitrate loop for a batch of frames, one call -------------------------------------------------
static size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

extern "C" size_t opusgpu_silk_encode_frames_cbr_workspace_bytes(int n)
{
    if (n <= 0) return 0;
    return 2 * al256((size_t)n * sizeof(opusgpu_nsq_state)) + 2 * al256((size_t)n * sizeof(opusgpu_ec_state)) + al256((size_t)n * sizeof(int)) + 256;
}

extern "C" int opusgpu_silk_encode_frames_cbr_batch(const opusgpu_silk_chain_bufs *bufs, opusgpu_silk_rate_ctl *d_ctl, int fs_kHz, int nb_subfr,
                                                    int del_dec, int n, void *d_loop_workspace, size_t loop_workspace_bytes, int *passes,
                                                    void *stream)
{
    if (!bufs || n < 0) return OPUSGPU_BAD_ARG;
    if (passes) *passes = 0;
    if (n == 0) return OPUSGPU_OK;
    const opusgpu_silk_chain_bufs b = *bufs;
    if (!d_ctl || !b.bits_in || !b.ec_state || !b.bits_out || !b.nsq_state || !b.q_in || !b.q_out || !b.gains_out || !d_loop_workspace)
        return OPUSGPU_BAD_ARG;                                       // the loop measures the coder: it needs the entropy-coding stage
    if (loop_workspace_bytes < opusgpu_silk_encode_frames_cbr_workspace_bytes(n)) return OPUSGPU_BUFFER_TOO_SMALL;
    LoopWs w;
    char *p = (char *)d_loop_workspace;
    w.nsq_entry = (opusgpu_nsq_state *)p; p += al256((size_t)n * sizeof(opusgpu_nsq_state));
    w.nsq_low = (opusgpu_nsq_state *)p; p += al256((size_t)n * sizeof(opusgpu_nsq_state));
    w.ec_entry = (opusgpu_ec_state *)p; p += al256((size_t)n * sizeof(opusgpu_ec_state));
    w.ec_low = (opusgpu_ec_state *)p; p += al256((size_t)n * sizeof(opusgpu_ec_state));
    w.rows = (int *)p; p += al256((size_t)n * sizeof(int));
    w.count = (int *)p;
    hipStream_t s = (hipStream_t)stream;
    const size_t qs = del_dec ? sizeof(opusgpu_nsq_dd_in) : sizeof(opusgpu_nsq_in);
    int rc;
    hipLaunchKernelGGL(loop_entry_kernel, dim3(n), dim3(64), 0, s, b, w, n);
    if ((rc = opusgpu_check_launch()) != OPUSGPU_OK) return rc;
    if ((rc = opusgpu_silk_encode_frames_batch(bufs, fs_kHz, nb_subfr, del_dec, n, stream)) != OPUSGPU_OK) return rc;
    hipLaunchKernelGGL(loop_init_kernel, dim3((n + 63) / 64), dim3(64), 0, s, b, d_ctl, n);
    if ((rc = opusgpu_check_launch()) != OPUSGPU_OK) return rc;
    // iter 0 .. maxIter (6): at most seven coded passes per frame, each followed by a step; the only host read of an iteration is
    // the length of the list of frames that go again
    for (int it = 0; it < 8; it++) {
        if (hipMemsetAsync(w.count, 0, 2 * sizeof(int), s) != hipSuccess) return OPUSGPU_INTERNAL_ERROR;
        if ((rc = opusgpu_silk_rate_control_batch(d_ctl, b.ec_state, n, stream)) != OPUSGPU_OK) return rc;
        hipLaunchKernelGGL(loop_act_kernel, dim3(n), dim3(64), 0, s, b, d_ctl, w, qs, n);
        if ((rc = opusgpu_check_launch()) != OPUSGPU_OK) return rc;
        int cnt[2] = {0, 0};
        if (hipMemcpyAsync(cnt, w.count, sizeof(cnt), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
            return OPUSGPU_INTERNAL_ERROR;
        if (cnt[1]) return OPUSGPU_BAD_ARG;                          // a rate-control record failed its checks (status in the record)
        const int m = cnt[0];
        if (m == 0) return OPUSGPU_OK;
        if (passes) *passes = it + 1;
        if (del_dec) rc = opusgpu_silk_nsq_del_dec_rows((const opusgpu_nsq_dd_in *)b.q_in, b.nsq_state, (opusgpu_nsq_dd_out *)b.q_out, w.rows, m, b.workspace, s);
        else rc = opusgpu_silk_nsq_rows((const opusgpu_nsq_in *)b.q_in, b.nsq_state, (opusgpu_nsq_out *)b.q_out, w.rows, m, b.workspace, s);
        if (rc != OPUSGPU_OK) return rc;
        hipLaunchKernelGGL(after_quantiser_rows_kernel, dim3(m), dim3(64), 0, s, b, del_dec, w.rows, m);
        if ((rc = opusgpu_check_launch()) != OPUSGPU_OK) return rc;
        if ((rc = opusgpu_silk_encode_bits_rows(b.bits_in, b.ec_state, b.bits_out, w.rows, m, s)) != OPUSGPU_OK) return rc;
    }
    return OPUSGPU_INTERNAL_ERROR;                                   // frames still asking for a pass after maxIter: cannot happen (:297)
}

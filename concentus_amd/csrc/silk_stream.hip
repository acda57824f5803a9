// silk_stream.hip -- streams mode of the SILK frame chain: what silk_encode_frame_FIX carries from one frame of a stream to the
// next (opus-fix/silk/fixed/encode_frame_FIX.c:128, :145, :427, :437-441; the psEnc fields its analysis calls leave behind), kept in
// a device record per stream, so that S streams x T frames run through opusgpu_silk_encode_frames(_cbr)_batch with nothing coming back
// to the host between frames. The per-frame INPUT of the function -- the frame's samples after the encoder's own input filters, the
// VAD results, SNR_dB_Q7, maxBits, condCoding: computed outside silk_encode_frame_FIX -- stays the caller's.
//   carry_in   stream record + the frame's samples -> the carried fields of frame t's records (every other field is the caller's or
//              is filled inside the frame by silk_chain.hip)
//   carry_out  frame t's outputs -> stream record
// The field list is pinned on the unmodified reference by tests/test_silk_stream_cpu.py. No arithmetic happens in this file.
#include <hip/hip_runtime.h>
#include "opusgpu_internal.h"
#include "../../include/opusgpu_silk.h"

namespace {

__device__ __forceinline__ opusgpu_nsq_in &qrec(void *q, size_t stride, int r) { return *(opusgpu_nsq_in *)((char *)q + stride * (size_t)r); }

// one 64-thread workgroup per stream
__global__ __launch_bounds__(64) void stream_carry_in_kernel(const opusgpu_silk_stream *__restrict__ streams, const int16_t *__restrict__ input,
                                                             opusgpu_silk_chain_bufs b, size_t qs, int fs_kHz, int nb_subfr, int n)
{
    const int r = blockIdx.x;
    if (r >= n) return;
    const int fl = 5 * fs_kHz * nb_subfr, ltp = 20 * fs_kHz, la_s = 5 * fs_kHz, la_p = 2 * fs_kHz;
    const opusgpu_silk_stream &st = streams[r];
    const int16_t *in = input + (size_t)r * OPUSGPU_SILK_MAX_FRAME;
    opusgpu_find_pitch_lags_in &p = const_cast<opusgpu_find_pitch_lags_in &>(b.pitch_in[r]);
    // x_buf of this frame (encode_frame_FIX.c:145): [ what the previous frame left: ltp + la_shape samples | the new frame ]
    for (int k = threadIdx.x; k < ltp + la_s + fl; k += 64) {
        const int16_t v = k < ltp + la_s ? st.x_buf[k] : in[k - (ltp + la_s)];
        if (k < ltp + fl + la_p) p.x_buf[k] = v;                                   // x[-ltp .. fl + la_pitch)
        if (k >= ltp - la_s) b.shape_in[r].x[k - (ltp - la_s)] = v;                // x[-la_shape .. fl + la_shape)
        if (k < ltp + fl) b.fpc_in[r].x[k] = v;                                    // x[-ltp .. fl)
        if (k >= ltp && k < ltp + fl) b.prefilter_in[r].x[k - ltp] = v;            // x[0 .. fl)
    }
    if (threadIdx.x < OPUSGPU_SILK_MAX_ORDER) b.fpc_in[r].prev_NLSFq_Q15[threadIdx.x] = st.prev_NLSFq_Q15[threadIdx.x];
    if (threadIdx.x == 0) {
        p.prevLag = st.prevLag; p.prevSignalType = st.prevSignalType; p.first_frame_after_reset = st.first_frame_after_reset;
        p.LTPCorr_Q15 = st.LTPCorr_Q15;
        b.shape_in[r].HarmBoost_smth_Q16 = st.HarmBoost_smth_Q16; b.shape_in[r].HarmShapeGain_smth_Q16 = st.HarmShapeGain_smth_Q16;
        b.shape_in[r].Tilt_smth_Q16 = st.Tilt_smth_Q16;
        b.fpc_in[r].first_frame_after_reset = st.first_frame_after_reset; b.fpc_in[r].sum_log_gain_Q7 = st.sum_log_gain_Q7;
        b.gains_in[r].LastGainIndex = st.LastGainIndex;
        qrec(b.q_in, qs, r).Seed = st.frameCounter & 3;                            // :128
        if (b.bits_in) {
            b.bits_in[r].Seed = st.frameCounter & 3;
            b.bits_in[r].ec_prevSignalType = st.ec_prevSignalType; b.bits_in[r].ec_prevLagIndex = st.ec_prevLagIndex;
        }
    }
}

__global__ __launch_bounds__(64) void stream_carry_out_kernel(opusgpu_silk_stream *__restrict__ streams, opusgpu_silk_chain_bufs b,
                                                              const opusgpu_silk_rate_ctl *__restrict__ ctl, int fs_kHz, int nb_subfr, int n)
{
    const int r = blockIdx.x;
    if (r >= n) return;
    const int fl = 5 * fs_kHz * nb_subfr, ltp = 20 * fs_kHz, la_s = 5 * fs_kHz, la_p = 2 * fs_kHz;
    opusgpu_silk_stream &st = streams[r];
    const opusgpu_find_pitch_lags_in &p = b.pitch_in[r];
    // x_buf moves up by one frame (:427): sample fl + k of this frame's buffer, found in the pitch record or, past its end, in the
    // shaping record
    for (int k = threadIdx.x; k < ltp + la_s; k += 64) {
        const int s = fl + k;
        st.x_buf[k] = s < ltp + fl + la_p ? p.x_buf[s] : b.shape_in[r].x[s - (ltp - la_s)];
    }
    if (threadIdx.x < OPUSGPU_SILK_MAX_ORDER) st.prev_NLSFq_Q15[threadIdx.x] = b.fpc_out[r].NLSF_Q15[threadIdx.x];
    if (threadIdx.x == 0) {
        const opusgpu_find_pitch_lags_out &po = b.pitch_out[r];
        st.prevLag = po.pitchL[nb_subfr - 1];                                       // :437
        st.prevSignalType = po.signalType;                                          // :438
        st.first_frame_after_reset = 0;                                             // :441
        st.LTPCorr_Q15 = po.LTPCorr_Q15;
        st.HarmBoost_smth_Q16 = b.shape_out[r].HarmBoost_smth_Q16; st.HarmShapeGain_smth_Q16 = b.shape_out[r].HarmShapeGain_smth_Q16;
        st.Tilt_smth_Q16 = b.shape_out[r].Tilt_smth_Q16;
        st.sum_log_gain_Q7 = b.fpc_out[r].sum_log_gain_Q7;
        st.LastGainIndex = ctl ? ctl[r].LastGainIndex : b.gains_out[r].LastGainIndex;
        if (b.bits_out) { st.ec_prevSignalType = b.bits_out[r].ec_prevSignalType; st.ec_prevLagIndex = b.bits_out[r].ec_prevLagIndex; }
        st.frameCounter = st.frameCounter + 1;
    }
}

bool geometry_ok(int fs_kHz, int nb_subfr) { return (fs_kHz == 8 || fs_kHz == 16) && (nb_subfr == 2 || nb_subfr == 4); }

}  // namespace

extern "C" int opusgpu_silk_stream_carry_in(const opusgpu_silk_stream *d_streams, const int16_t *d_input, const opusgpu_silk_chain_bufs *bufs,
                                            int fs_kHz, int nb_subfr, int del_dec, int n, void *stream)
{
    if (!bufs || n < 0) return OPUSGPU_BAD_ARG;
    if (n == 0) return OPUSGPU_OK;
    const opusgpu_silk_chain_bufs b = *bufs;
    if (!d_streams || !d_input || !b.pitch_in || !b.shape_in || !b.fpc_in || !b.gains_in || !b.prefilter_in || !b.q_in || !geometry_ok(fs_kHz, nb_subfr))
        return OPUSGPU_BAD_ARG;
    const size_t qs = del_dec ? sizeof(opusgpu_nsq_dd_in) : sizeof(opusgpu_nsq_in);
    hipLaunchKernelGGL(stream_carry_in_kernel, dim3(n), dim3(64), 0, (hipStream_t)stream, d_streams, d_input, b, qs, fs_kHz, nb_subfr, n);
    return opusgpu_check_launch();
}

extern "C" int opusgpu_silk_stream_carry_out(opusgpu_silk_stream *d_streams, const opusgpu_silk_chain_bufs *bufs, const opusgpu_silk_rate_ctl *d_ctl,
                                             int fs_kHz, int nb_subfr, int n, void *stream)
{
    if (!bufs || n < 0) return OPUSGPU_BAD_ARG;
    if (n == 0) return OPUSGPU_OK;
    const opusgpu_silk_chain_bufs b = *bufs;
    if (!d_streams || !b.pitch_in || !b.pitch_out || !b.shape_in || !b.shape_out || !b.fpc_out || !b.gains_out || !geometry_ok(fs_kHz, nb_subfr))
        return OPUSGPU_BAD_ARG;
    hipLaunchKernelGGL(stream_carry_out_kernel, dim3(n), dim3(64), 0, (hipStream_t)stream, d_streams, b, d_ctl, fs_kHz, nb_subfr, n);
    return opusgpu_check_launch();
}

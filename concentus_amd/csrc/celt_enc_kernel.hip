// celt_enc_kernel.hip -- batched Opus CELT-only encode for gfx950 (BASELINE config #3).
//
// One 64-lane wavefront encodes one 20 ms frame (workgroup = one wave), in two kernels so that each
// phase keeps only its own working set in LDS (see FrontLds / BackLds in celt_enc_front.h):
//   celt_front_kernel  PCM -> FrameMid   (dc_reject, pre-emphasis, pitch pre-filter, transient analysis,
//                                         MDCT, band energies, normalisation): ~23 KB LDS, 6 waves/CU
//   celt_back_kernel   FrameMid -> packet (TF, coarse/fine energy, allocation, PVQ, range coder):
//                                         ~9.6 KB LDS, 16 waves/CU -- the serial, latency-bound 70 %
// FrameMid records live in a caller-provided HBM workspace; batches larger than the workspace are
// processed in chunks on the same stream. Frames are independent units (streams advance one frame per
// call), so the batch shards across workgroups, CUs and GPUs with no communication.
//
// Replaces, for 48 kHz / 20 ms / restricted-lowdelay / fullband: opus_encode() (opus-fix/src/opus_encoder.c:2007)
// -> opus_encode_native (:938) -> celt_encode_with_ec (opus-fix/celt/celt_encoder.c:1379).
#include <stdlib.h>
#include "celt_enc.h"
#include "opusgpu_internal.h"
extern "C" void opusgpu_launch_dc_reject(const void *states, const int16_t *pcm, void *mid, int n, hipStream_t s);
extern "C" void opusgpu_launch_transient(void *mid, const int32_t *in_ws, int n, hipStream_t s);
extern "C" void opusgpu_launch_back_lane(const opusgpu_celt_config *cfg, void *states, const void *mid, unsigned char *out,
                                         int out_stride, int32_t *out_len, uint32_t *out_rng, int n, hipStream_t s);

namespace ca {

__global__ __launch_bounds__(64, 2) void celt_front_kernel(opusgpu_celt_config cfg, opusgpu_celt_state *states,
                                                           const i16 *__restrict__ pcm, FrameMid *__restrict__ mid, int nframes)
{
    __shared__ FrontLds F;
    for (int n = blockIdx.x; n < nframes; n += gridDim.x) {
        opusgpu_celt_state *st = states ? states + n : nullptr;
        celt_encode_front(F, cfg, st, st, pcm + (size_t)n * FRAME * cfg.channels, mid + n);
        wave_sync();
    }
}

// Split front phase (default pipeline): phase 1 = rate bookkeeping .. pitch pre-filter, phase 2 = MDCT ..
// normalisation; the serial dc_reject and transient stages run in celt_stage_kernels.hip in between.
__global__ __launch_bounds__(64, 3) void celt_front1_kernel(opusgpu_celt_config cfg, opusgpu_celt_state *states,
                                                            FrameMid *__restrict__ mid, i32 *__restrict__ in_ws, int nframes)
{
    __shared__ Front1Lds F;
    for (int n = blockIdx.x; n < nframes; n += gridDim.x) {
        opusgpu_celt_state *st = states ? states + n : nullptr;
        i32 *in = in_ws + (size_t)n * 2 * (FRAME + OVL);
        if (lane() == 0) F.in_g = in;
        wave_sync();
        celt_encode_front_phase<1>(F, cfg, st, st, nullptr, mid + n, nullptr, in);
        wave_sync();
    }
}

// (64, 4): the one-channel working set (8.7 KB of LDS) admits 18 wavefronts per CU; the register budget is set for 16
__global__ __launch_bounds__(64, 4) void celt_front2_kernel(opusgpu_celt_config cfg, FrameMid *__restrict__ mid,
                                                            i32 *__restrict__ in_ws, int nframes)
{
    __shared__ Front2Lds F;
    for (int n = blockIdx.x; n < nframes; n += gridDim.x) {
        i32 *in = in_ws + (size_t)n * 2 * (FRAME + OVL);
        if (lane() == 0) { F.in_g = in; F.x_g = mid[n].X; }
        wave_sync();
        celt_encode_front_phase<2>(F, cfg, nullptr, nullptr, nullptr, mid + n, nullptr, in);
        wave_sync();
    }
}

__global__ __launch_bounds__(64, 4) void celt_back_kernel(opusgpu_celt_config cfg, opusgpu_celt_state *states,
                                                          const FrameMid *__restrict__ mid, u8 *__restrict__ out, int out_stride,
                                                          int *__restrict__ out_len, u32 *__restrict__ out_rng, int nframes)
{
    __shared__ BackLds F;
    for (int n = blockIdx.x; n < nframes; n += gridDim.x) {
        opusgpu_celt_state *st = states ? states + n : nullptr;
        FrameResult r = celt_encode_back(F, cfg, mid + n, st, out + (size_t)n * out_stride);
        if (lane() == 0) {
            out_len[n] = r.bytes;
            out_rng[n] = r.final_range;
        }
        wave_sync();
    }
}

// fresh encoder state (opus_encoder_create + OPUS_RESET_STATE, celt_encoder.c:2443-2462)
__global__ void celt_state_init_kernel(opusgpu_celt_state *states, int n)
{
    int i = blockIdx.x;
    if (i >= n) return;
    u32 *w = reinterpret_cast<u32 *>(&states[i]);
    for (int k = threadIdx.x; k < (int)(sizeof(opusgpu_celt_state) / 4); k += blockDim.x) w[k] = 0;
    __syncthreads();
    if (threadIdx.x == 0) {
        states[i].spread_decision = SPREAD_NORMAL;
        states[i].delayedIntra = 1;
        states[i].tonal_average = 256;
    }
    for (int k = threadIdx.x; k < 2 * NB; k += blockDim.x) {
        states[i].oldLogE[k] = -28672;
        states[i].oldLogE2[k] = -28672;
    }
}

}  // namespace ca

using namespace ca;

static int config_ok(const opusgpu_celt_config *c)
{
    if (!c) return OPUSGPU_BAD_ARG;
    if (c->channels != 2 && c->channels != 1) return OPUSGPU_BAD_ARG;
    if (c->complexity < 0 || c->complexity > 10 || c->loss_rate < 0 || c->loss_rate > 100) return OPUSGPU_BAD_ARG;
    if (c->lsb_depth < 8 || c->lsb_depth > 24 || c->max_data_bytes <= 0) return OPUSGPU_BAD_ARG;
    if (c->bitrate <= 0) return OPUSGPU_BAD_ARG;
    // Only the operating region where the Opus layer's automatic decisions are constant is implemented:
    // stereo kept stereo (equiv_rate > stereo threshold 30 kb/s +- 1 kb/s hysteresis, opus_encoder.c:1121-1131)
    // and bandwidth stays FULLBAND (stereo music/voice thresholds <= 30 kb/s, :1263-1305).
    if (c->channels != 2 || c->bitrate < 32000) return OPUSGPU_UNIMPLEMENTED;
    // CBR: the packet size caps the rate the decisions see (cbrBytes, opus_encoder.c:1054-1061)
    if (!c->vbr) {
        int maxb = c->max_data_bytes < 1276 ? c->max_data_bytes : 1276;
        int cbr = (3 * c->bitrate / 8 + 75) / 150;
        if ((cbr < maxb ? cbr : maxb) * 400 < 32000) return OPUSGPU_UNIMPLEMENTED;
    }
    // PLC-frame corner (opus_encoder.c:1056-1084) and too-small buffers
    if (c->max_data_bytes < 3 || c->bitrate > 1276 * 400) return OPUSGPU_UNIMPLEMENTED;     // 510 400 = OPUS_BITRATE_MAX into a 1276-byte buffer
    return OPUSGPU_OK;
}

extern "C" int opusgpu_celt_state_size(void) { return (int)sizeof(opusgpu_celt_state); }

extern "C" int opusgpu_celt_state_init(void *d_states, int n_streams, void *stream)
{
    if (n_streams < 0) return OPUSGPU_BAD_ARG;
    if (n_streams == 0) return OPUSGPU_OK;
    if (!d_states) return OPUSGPU_BAD_ARG;
    hipLaunchKernelGGL(celt_state_init_kernel, dim3(n_streams), dim3(256), 0, (hipStream_t)stream,
                       (opusgpu_celt_state *)d_states, n_streams);
    return opusgpu_check_launch();
}

// per frame in flight: the FrameMid record + the [2][1080] int32 time signal handed from front phase 1 to 2
static const size_t WS_IN_BYTES = 2 * (FRAME + OVL) * sizeof(i32);
static const size_t WS_FRAME_BYTES = sizeof(FrameMid) + WS_IN_BYTES;

extern "C" size_t opusgpu_encode_workspace_bytes(int n_frames)
{
    return n_frames <= 0 ? 0 : (size_t)n_frames * WS_FRAME_BYTES;
}

extern "C" int opusgpu_encode_batch(const opusgpu_celt_config *cfg, void *d_states, const int16_t *d_pcm,
                                    unsigned char *d_out, int out_stride, int32_t *d_out_len, uint32_t *d_out_rng,
                                    int n_frames, void *d_workspace, size_t workspace_bytes, void *stream)
{
    int rc = config_ok(cfg);
    if (rc != OPUSGPU_OK) return rc;
    if (n_frames < 0) return OPUSGPU_BAD_ARG;
    if (n_frames == 0) return OPUSGPU_OK;
    if (!d_pcm || !d_out || !d_out_len || !d_out_rng || !d_workspace) return OPUSGPU_BAD_ARG;
    int maxbytes = cfg->max_data_bytes < 1276 ? cfg->max_data_bytes : 1276;
    if (out_stride < ((maxbytes + 3) & ~3) || (out_stride & 3)) return OPUSGPU_BUFFER_TOO_SMALL;
    size_t chunk = workspace_bytes / WS_FRAME_BYTES;
    if (chunk == 0) return OPUSGPU_BUFFER_TOO_SMALL;
    const int cus = opusgpu_num_cus();
    opusgpu_celt_state *st = (opusgpu_celt_state *)d_states;
    FrameMid *mid = (FrameMid *)d_workspace;
    i32 *in_ws = (i32 *)((char *)d_workspace + chunk * sizeof(FrameMid));
    hipStream_t s = (hipStream_t)stream;
    // default: one lane per frame for the serial back phase; OPUSGPU_BACK_WAVE=1 selects the one-wave-per-frame kernel
    // (kept for the stage-stamp diagnostics and as a cross-check); OPUSGPU_FRONT_FUSED=1: the whole front phase in one
    // wave-per-frame kernel (cross-check / diagnostics). Read once per call, not per chunk.
    const bool lane_back = getenv("OPUSGPU_BACK_WAVE") == nullptr;
    const bool fused_front = getenv("OPUSGPU_FRONT_FUSED") != nullptr;
    for (size_t first = 0; first < (size_t)n_frames; first += chunk) {
        int n = (int)(((size_t)n_frames - first) < chunk ? ((size_t)n_frames - first) : chunk);
        int g1 = n < cus * 7 ? n : cus * 7;          // 22.8 KB LDS per workgroup -> 7 resident per CU (measured best)
        int g2 = n < cus * 16 ? n : cus * 16;        // ~9.6 KB LDS per workgroup -> 16 resident per CU
        int slot;
        if (fused_front) {
            slot = opusgpu_timing_begin(OPUSGPU_KERNEL_CELT_FRONT, s);
            hipLaunchKernelGGL(celt_front_kernel, dim3(g1), dim3(64), 0, s, *cfg, st ? st + first : nullptr,
                               d_pcm + first * FRAME * cfg->channels, mid, n);
            opusgpu_timing_end(slot, s);
        } else {
            slot = opusgpu_timing_begin(OPUSGPU_KERNEL_CELT_DC_REJECT, s);
            opusgpu_launch_dc_reject(st ? st + first : nullptr, d_pcm + first * FRAME * cfg->channels, mid, n, s);
            opusgpu_timing_end(slot, s);
            // persistent grids sized to what a CU holds: 13.9 KB of LDS per workgroup (11) / 8.7 KB at <= 128 VGPRs (16)
            static const int f2_waves = getenv("OPUSGPU_FRONT2_WAVES") ? atoi(getenv("OPUSGPU_FRONT2_WAVES")) : 16;
            const int gf1 = n < cus * 11 ? n : cus * 11, gf2 = n < cus * f2_waves ? n : cus * f2_waves;
            slot = opusgpu_timing_begin(OPUSGPU_KERNEL_CELT_FRONT1, s);
            hipLaunchKernelGGL(celt_front1_kernel, dim3(gf1), dim3(64), 0, s, *cfg, st ? st + first : nullptr, mid, in_ws, n);
            opusgpu_timing_end(slot, s);
            slot = opusgpu_timing_begin(OPUSGPU_KERNEL_CELT_TRANSIENT, s);
            opusgpu_launch_transient(mid, in_ws, n, s);
            opusgpu_timing_end(slot, s);
            slot = opusgpu_timing_begin(OPUSGPU_KERNEL_CELT_FRONT2, s);
            hipLaunchKernelGGL(celt_front2_kernel, dim3(gf2), dim3(64), 0, s, *cfg, mid, in_ws, n);
            opusgpu_timing_end(slot, s);
        }
        slot = opusgpu_timing_begin(lane_back ? OPUSGPU_KERNEL_CELT_BACK_LANE : OPUSGPU_KERNEL_CELT_BACK, s);
        if (lane_back)
            opusgpu_launch_back_lane(cfg, st ? st + first : nullptr, mid, d_out + first * (size_t)out_stride, out_stride,
                                     d_out_len + first, d_out_rng + first, n, s);
        else
        hipLaunchKernelGGL(celt_back_kernel, dim3(g2), dim3(64), 0, s, *cfg, st ? st + first : nullptr, mid,
                           d_out + first * (size_t)out_stride, out_stride, d_out_len + first, d_out_rng + first, n);
        opusgpu_timing_end(slot, s);
    }
    return opusgpu_check_launch();
}

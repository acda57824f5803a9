// celt_enc_kernel.hip -- batched Opus CELT-only encode for gfx950 (BASELINE config #3).
//
// One 64-lane wavefront encodes one 20 ms frame end to end (PCM -> packet): workgroup = one wave, its
// whole working set (FrameLds, ~26 KB) in LDS, grid-stride over the frames of the batch. Frames are
// independent units (streams advance one frame per launch), so the batch shards across workgroups,
// CUs and GPUs with no communication.
//
// Replaces, for 48 kHz / 20 ms / restricted-lowdelay / fullband: opus_encode() (opus-fix/src/opus_encoder.c:2007)
// -> opus_encode_native (:938) -> celt_encode_with_ec (opus-fix/celt/celt_encoder.c:1379).
#include "celt_enc.h"
#include "opusgpu_internal.h"

namespace ca {

__global__ __launch_bounds__(64, 2) void celt_encode_kernel(opusgpu_celt_config cfg, opusgpu_celt_state *states,
                                                         const i16 *__restrict__ pcm, u8 *__restrict__ out, int out_stride,
                                                         int *__restrict__ out_len, u32 *__restrict__ out_rng, int nframes)
{
    __shared__ FrameLds F;
    for (int n = blockIdx.x; n < nframes; n += gridDim.x) {
        opusgpu_celt_state *st = states ? states + n : nullptr;
        FrameResult r = celt_encode_frame(F, cfg, st, st, pcm + (size_t)n * FRAME * cfg.channels, out + (size_t)n * out_stride);
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
    // PLC-frame corner (opus_encoder.c:1056-1084) and too-small buffers
    if (c->max_data_bytes < 3 || c->bitrate > 510000) return OPUSGPU_UNIMPLEMENTED;
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

extern "C" int opusgpu_encode_batch(const opusgpu_celt_config *cfg, void *d_states, const int16_t *d_pcm,
                                    unsigned char *d_out, int out_stride, int32_t *d_out_len, uint32_t *d_out_rng,
                                    int n_frames, void *stream)
{
    int rc = config_ok(cfg);
    if (rc != OPUSGPU_OK) return rc;
    if (n_frames < 0) return OPUSGPU_BAD_ARG;
    if (n_frames == 0) return OPUSGPU_OK;
    if (!d_pcm || !d_out || !d_out_len || !d_out_rng) return OPUSGPU_BAD_ARG;
    int maxbytes = cfg->max_data_bytes < 1276 ? cfg->max_data_bytes : 1276;
    if (out_stride < ((maxbytes + 3) & ~3) || (out_stride & 3)) return OPUSGPU_BUFFER_TOO_SMALL;
    int cap = opusgpu_num_cus() * 6;               // ~26 KB LDS per workgroup -> 6 resident per CU
    int grid = n_frames < cap ? n_frames : cap;
    hipLaunchKernelGGL(celt_encode_kernel, dim3(grid), dim3(64), 0, (hipStream_t)stream, *cfg,
                       (opusgpu_celt_state *)d_states, d_pcm, d_out, out_stride, d_out_len, d_out_rng, n_frames);
    return opusgpu_check_launch();
}

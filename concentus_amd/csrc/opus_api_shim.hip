// opus_api_shim.hip -- the libopus single-stream encoder API as a batch of one (plumbing, BASELINE config #1).
//
// Same verbs and argument meaning as opus_encoder_create / opus_encoder_ctl / opus_encode /
// opus_encoder_destroy (opus-fix/include/opus.h:164-263, src/opus_encoder.c:482,2031,2007,2491), host
// pointers in and out. Each opusgpu_encode() is one opusgpu_encode_batch() of one frame bracketed by two
// small copies, so it is latency-bound by construction; throughput comes from the batch entry point.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include "opusgpu_internal.h"

struct OpusGpuEncoder {
    opusgpu_celt_config cfg;
    void *d_state, *d_ws;
    int16_t *d_pcm;
    unsigned char *d_out;
    int32_t *d_len;
    uint32_t *d_rng;
    size_t ws_bytes;
    uint32_t final_range;
    int bitrate_is_max;            // OPUS_SET_BITRATE(OPUS_BITRATE_MAX): resolved per opus_encode() call (src/opus_encoder.c:512-521)
    hipStream_t stream;
};

enum { OUT_STRIDE = 1276 };

static void free_all(OpusGpuEncoder *st)
{
    if (!st) return;
    if (st->d_state) (void)hipFree(st->d_state);
    if (st->d_ws) (void)hipFree(st->d_ws);
    if (st->d_pcm) (void)hipFree(st->d_pcm);
    if (st->d_out) (void)hipFree(st->d_out);
    if (st->d_len) (void)hipFree(st->d_len);
    if (st->d_rng) (void)hipFree(st->d_rng);
    if (st->stream) (void)hipStreamDestroy(st->stream);
    free(st);
}

extern "C" OpusGpuEncoder *opusgpu_encoder_create(int32_t Fs, int channels, int application, int *error)
{
    int err = OPUSGPU_OK;
    OpusGpuEncoder *st = nullptr;
    if ((Fs != 48000 && Fs != 24000 && Fs != 16000 && Fs != 12000 && Fs != 8000) || (channels != 1 && channels != 2) ||
        (application != 2048 && application != 2049 && application != 2051))
        err = OPUSGPU_BAD_ARG;                                                   // src/opus_encoder.c:487-494
    else if (Fs != 48000 || application != 2051 /* OPUS_APPLICATION_RESTRICTED_LOWDELAY */ || channels != 2)
        err = OPUSGPU_UNIMPLEMENTED;
    if (err == OPUSGPU_OK) {
        st = (OpusGpuEncoder *)calloc(1, sizeof(OpusGpuEncoder));
        if (!st) err = OPUSGPU_ALLOC_FAIL;
    }
    if (err == OPUSGPU_OK) {
        // opus_encoder_init defaults (src/opus_encoder.c:229-252): VBR on, constrained, auto bitrate, complexity 9, 24 bit
        st->cfg.channels = channels;
        st->cfg.bitrate = 3000 + Fs * channels;
        st->cfg.vbr = 1;
        st->cfg.constrained_vbr = 1;
        st->cfg.complexity = 9;
        st->cfg.lsb_depth = 24;
        st->cfg.loss_rate = 0;
        st->cfg.max_data_bytes = 1276;
        st->ws_bytes = opusgpu_encode_workspace_bytes(1);
        if (hipStreamCreate(&st->stream) != hipSuccess || hipMalloc(&st->d_state, (size_t)opusgpu_celt_state_size()) != hipSuccess ||
            hipMalloc(&st->d_ws, st->ws_bytes) != hipSuccess || hipMalloc((void **)&st->d_pcm, 960 * 2 * sizeof(int16_t)) != hipSuccess ||
            hipMalloc((void **)&st->d_out, OUT_STRIDE) != hipSuccess || hipMalloc((void **)&st->d_len, 4) != hipSuccess ||
            hipMalloc((void **)&st->d_rng, 4) != hipSuccess)
            err = OPUSGPU_ALLOC_FAIL;
        else
            err = opusgpu_celt_state_init(st->d_state, 1, st->stream);
        if (err != OPUSGPU_OK) { free_all(st); st = nullptr; }
    }
    if (error) *error = err;
    return st;
}

extern "C" void opusgpu_encoder_destroy(OpusGpuEncoder *st) { free_all(st); }

// The requests opus_demo issues (src/opus_demo.c:531-543) plus OPUS_GET_FINAL_RANGE and OPUS_RESET_STATE.
extern "C" int opusgpu_encoder_ctl(OpusGpuEncoder *st, int request, ...)
{
    if (!st) return OPUSGPU_BAD_ARG;
    va_list ap;
    va_start(ap, request);
    int ret = OPUSGPU_OK;
    switch (request) {
    case 4002: {                                                                  // OPUS_SET_BITRATE (:2066-2083)
        int32_t v = va_arg(ap, int32_t);
        st->bitrate_is_max = v == -1;                                             // OPUS_BITRATE_MAX
        if (v == -1000) v = 3000 + 48000 * st->cfg.channels;                      // OPUS_AUTO
        else if (v == -1) v = 1276 * 400;                                         // placeholder; see opusgpu_encode
        else if (v <= 0) { ret = OPUSGPU_BAD_ARG; break; }
        else if (v <= 500) v = 500;
        else if (v > 300000 * st->cfg.channels) v = 300000 * st->cfg.channels;
        st->cfg.bitrate = v;
        break;
    }
    case 4006: st->cfg.vbr = va_arg(ap, int32_t) != 0; break;                    // OPUS_SET_VBR
    case 4020: st->cfg.constrained_vbr = va_arg(ap, int32_t) != 0; break;        // OPUS_SET_VBR_CONSTRAINT
    case 4010: {                                                                  // OPUS_SET_COMPLEXITY
        int32_t v = va_arg(ap, int32_t);
        if (v < 0 || v > 10) ret = OPUSGPU_BAD_ARG; else st->cfg.complexity = v;
        break;
    }
    case 4014: {                                                                  // OPUS_SET_PACKET_LOSS_PERC
        int32_t v = va_arg(ap, int32_t);
        if (v < 0 || v > 100) ret = OPUSGPU_BAD_ARG; else st->cfg.loss_rate = v;
        break;
    }
    case 4036: {                                                                  // OPUS_SET_LSB_DEPTH
        int32_t v = va_arg(ap, int32_t);
        if (v < 8 || v > 24) ret = OPUSGPU_BAD_ARG; else st->cfg.lsb_depth = v;
        break;
    }
    case 4008: case 4022:                                                         // OPUS_SET_BANDWIDTH / _FORCE_CHANNELS
        if (va_arg(ap, int32_t) != -1000) ret = OPUSGPU_UNIMPLEMENTED;
        break;
    case 4012: case 4016:                                                         // OPUS_SET_INBAND_FEC / _DTX (SILK features)
        if (va_arg(ap, int32_t) != 0) ret = OPUSGPU_UNIMPLEMENTED;
        break;
    case 4040:                                                                    // OPUS_SET_EXPERT_FRAME_DURATION
        if (va_arg(ap, int32_t) != 5000) ret = OPUSGPU_UNIMPLEMENTED;             // OPUS_FRAMESIZE_ARG
        break;
    case 4031: {                                                                  // OPUS_GET_FINAL_RANGE
        uint32_t *p = va_arg(ap, uint32_t *);
        if (!p) ret = OPUSGPU_BAD_ARG; else *p = st->final_range;
        break;
    }
    // getters (src/opus_encoder.c:2084-2431): every one takes an opus_int32 *
    case 4001: case 4003: case 4007: case 4009: case 4011: case 4013: case 4015: case 4017: case 4021: case 4023:
    case 4027: case 4029: case 4037: case 4041: {
        int32_t *p = va_arg(ap, int32_t *);
        if (!p) { ret = OPUSGPU_BAD_ARG; break; }
        switch (request) {
        case 4001: *p = 2051; break;                                              // OPUS_GET_APPLICATION: RESTRICTED_LOWDELAY
        case 4003: *p = st->bitrate_is_max ? 1276 * 400 : st->cfg.bitrate; break; // OPUS_GET_BITRATE (:2086-2093)
        case 4007: *p = st->cfg.vbr; break;
        case 4009: *p = 1105; break;                                              // OPUS_GET_BANDWIDTH: OPUS_BANDWIDTH_FULLBAND
        case 4011: *p = st->cfg.complexity; break;
        case 4013: case 4017: *p = 0; break;                                      // in-band FEC / DTX: off
        case 4015: *p = st->cfg.loss_rate; break;
        case 4021: *p = st->cfg.constrained_vbr; break;
        case 4023: *p = -1000; break;                                             // OPUS_GET_FORCE_CHANNELS: OPUS_AUTO
        case 4027: *p = 48000 / 400; break;                                       // OPUS_GET_LOOKAHEAD, restricted-lowdelay (:2336-2338)
        case 4029: *p = 48000; break;
        case 4037: *p = st->cfg.lsb_depth; break;
        default: *p = 5000; break;                                                // OPUS_GET_EXPERT_FRAME_DURATION: OPUS_FRAMESIZE_ARG
        }
        break;
    }
    case 4028:                                                                    // OPUS_RESET_STATE
        ret = opusgpu_celt_state_init(st->d_state, 1, st->stream);
        st->final_range = 0;
        break;
    default: ret = OPUSGPU_UNIMPLEMENTED;
    }
    va_end(ap);
    return ret;
}

extern "C" int32_t opusgpu_encode(OpusGpuEncoder *st, const int16_t *pcm, int frame_size, unsigned char *data, int32_t max_data_bytes)
{
    if (!st || !pcm || !data || max_data_bytes <= 0) return OPUSGPU_BAD_ARG;
    if (frame_size != 960) return frame_size == 120 || frame_size == 240 || frame_size == 480 || frame_size == 1920 || frame_size == 2880
                                      ? OPUSGPU_UNIMPLEMENTED : OPUSGPU_BAD_ARG;   // 20 ms @ 48 kHz only
    st->cfg.max_data_bytes = max_data_bytes;
    // user_bitrate_to_bitrate (src/opus_encoder.c:512-521, called at :1050 with max_data_bytes = IMIN(1276, out_data_bytes)):
    // OPUS_BITRATE_MAX means "whatever fills this call's buffer" = max_data_bytes * 8 * Fs / frame_size
    if (st->bitrate_is_max) st->cfg.bitrate = (max_data_bytes < 1276 ? max_data_bytes : 1276) * 8 * (48000 / 960);
    if (hipMemcpyAsync(st->d_pcm, pcm, 960 * 2 * sizeof(int16_t), hipMemcpyHostToDevice, st->stream) != hipSuccess) return OPUSGPU_INTERNAL_ERROR;
    int rc = opusgpu_encode_batch(&st->cfg, st->d_state, st->d_pcm, st->d_out, OUT_STRIDE, st->d_len, st->d_rng, 1, st->d_ws,
                                  st->ws_bytes, st->stream);
    if (rc < 0) return rc;
    int32_t len = 0;
    unsigned char host[OUT_STRIDE];
    if (hipMemcpyAsync(&len, st->d_len, 4, hipMemcpyDeviceToHost, st->stream) != hipSuccess ||
        hipMemcpyAsync(&st->final_range, st->d_rng, 4, hipMemcpyDeviceToHost, st->stream) != hipSuccess ||
        hipMemcpyAsync(host, st->d_out, OUT_STRIDE, hipMemcpyDeviceToHost, st->stream) != hipSuccess ||
        hipStreamSynchronize(st->stream) != hipSuccess)
        return OPUSGPU_INTERNAL_ERROR;
    if (len > 0) memcpy(data, host, (size_t)(len < max_data_bytes ? len : max_data_bytes));
    return len;
}

// ---- decoder side: opus_decoder_create / opus_decode / opus_decoder_ctl / opus_decoder_destroy as a batch of one --
struct OpusGpuDecoder {
    void *d_state;
    unsigned char *d_pkt;
    int32_t *d_len, *d_ret;
    uint32_t *d_rng;
    int16_t *d_pcm;
    uint32_t final_range;
    int last_packet_duration;
    hipStream_t stream;
};

static void free_all(OpusGpuDecoder *st)
{
    if (!st) return;
    if (st->d_state) (void)hipFree(st->d_state);
    if (st->d_pkt) (void)hipFree(st->d_pkt);
    if (st->d_len) (void)hipFree(st->d_len);
    if (st->d_ret) (void)hipFree(st->d_ret);
    if (st->d_rng) (void)hipFree(st->d_rng);
    if (st->d_pcm) (void)hipFree(st->d_pcm);
    if (st->stream) (void)hipStreamDestroy(st->stream);
    free(st);
}

extern "C" OpusGpuDecoder *opusgpu_decoder_create(int32_t Fs, int channels, int *error)
{
    int err = OPUSGPU_OK;
    OpusGpuDecoder *st = nullptr;
    if ((Fs != 48000 && Fs != 24000 && Fs != 16000 && Fs != 12000 && Fs != 8000) || (channels != 1 && channels != 2))
        err = OPUSGPU_BAD_ARG;                                                   // src/opus_decoder.c:127-133
    else if (Fs != 48000 || channels != 2)
        err = OPUSGPU_UNIMPLEMENTED;
    if (err == OPUSGPU_OK) {
        st = (OpusGpuDecoder *)calloc(1, sizeof(OpusGpuDecoder));
        if (!st) err = OPUSGPU_ALLOC_FAIL;
    }
    if (err == OPUSGPU_OK) {
        if (hipStreamCreate(&st->stream) != hipSuccess || hipMalloc(&st->d_state, (size_t)opusgpu_celt_dec_state_size()) != hipSuccess ||
            hipMalloc((void **)&st->d_pkt, 1280) != hipSuccess || hipMalloc((void **)&st->d_len, 4) != hipSuccess ||
            hipMalloc((void **)&st->d_ret, 4) != hipSuccess || hipMalloc((void **)&st->d_rng, 4) != hipSuccess ||
            hipMalloc((void **)&st->d_pcm, 960 * 2 * sizeof(int16_t)) != hipSuccess)
            err = OPUSGPU_ALLOC_FAIL;
        else
            err = opusgpu_celt_dec_state_init(st->d_state, 1, st->stream);
        if (err != OPUSGPU_OK) { free_all(st); st = nullptr; }
    }
    if (error) *error = err;
    return st;
}

extern "C" void opusgpu_decoder_destroy(OpusGpuDecoder *st) { free_all(st); }

extern "C" int opusgpu_decoder_ctl(OpusGpuDecoder *st, int request, ...)
{
    if (!st) return OPUSGPU_BAD_ARG;
    va_list ap;
    va_start(ap, request);
    int ret = OPUSGPU_OK;
    switch (request) {
    case 4031: {                                                                  // OPUS_GET_FINAL_RANGE
        uint32_t *p = va_arg(ap, uint32_t *);
        if (!p) ret = OPUSGPU_BAD_ARG; else *p = st->final_range;
        break;
    }
    case 4009: case 4029: case 4039: case 4045: {                                 // getters (src/opus_decoder.c:828-938)
        int32_t *p = va_arg(ap, int32_t *);
        if (!p) ret = OPUSGPU_BAD_ARG;
        else *p = request == 4009 ? 1105 : request == 4029 ? 48000 : request == 4039 ? st->last_packet_duration : 0;
        break;
    }
    case 4034:                                                                    // OPUS_SET_GAIN: only 0 dB
        if (va_arg(ap, int32_t) != 0) ret = OPUSGPU_UNIMPLEMENTED;
        break;
    case 4028:                                                                    // OPUS_RESET_STATE
        ret = opusgpu_celt_dec_state_init(st->d_state, 1, st->stream);
        st->final_range = 0;
        st->last_packet_duration = 0;
        break;
    default: ret = OPUSGPU_UNIMPLEMENTED;
    }
    va_end(ap);
    return ret;
}

// opus_decode(st, data, len, pcm, frame_size, decode_fec): returns the number of decoded samples per channel
extern "C" int opusgpu_decode(OpusGpuDecoder *st, const unsigned char *data, int32_t len, int16_t *pcm, int frame_size, int decode_fec)
{
    if (!st || !pcm || frame_size <= 0 || len < 0) return OPUSGPU_BAD_ARG;
    if (!data || len == 0 || decode_fec) return OPUSGPU_UNIMPLEMENTED;           // packet loss concealment / FEC
    if (len > 1276) return OPUSGPU_INVALID_PACKET;
    if (frame_size < 960) return OPUSGPU_BUFFER_TOO_SMALL;
    int32_t ret = 0;
    if (hipMemcpyAsync(st->d_pkt, data, (size_t)len, hipMemcpyHostToDevice, st->stream) != hipSuccess ||
        hipMemcpyAsync(st->d_len, &len, 4, hipMemcpyHostToDevice, st->stream) != hipSuccess)
        return OPUSGPU_INTERNAL_ERROR;
    int rc = opusgpu_decode_batch(st->d_state, st->d_pkt, 1280, st->d_len, st->d_pcm, st->d_ret, st->d_rng, 1, st->stream);
    if (rc < 0) return rc;
    if (hipMemcpyAsync(&ret, st->d_ret, 4, hipMemcpyDeviceToHost, st->stream) != hipSuccess ||
        hipMemcpyAsync(&st->final_range, st->d_rng, 4, hipMemcpyDeviceToHost, st->stream) != hipSuccess ||
        hipStreamSynchronize(st->stream) != hipSuccess)
        return OPUSGPU_INTERNAL_ERROR;
    if (ret > 0) {
        st->last_packet_duration = ret;
        if (hipMemcpy(pcm, st->d_pcm, (size_t)ret * 2 * sizeof(int16_t), hipMemcpyDeviceToHost) != hipSuccess) return OPUSGPU_INTERNAL_ERROR;
    }
    return ret;
}

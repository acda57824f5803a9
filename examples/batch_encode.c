/* examples/batch_encode.c -- a C host for the BATCHED boundary (include/opusgpu.h), no C++ and no Python.
 *
 *   batch_encode <in.pcm> <out.bit> [bitrate [vbr [cvbr [complexity]]]]
 *
 * Reads 48 kHz stereo s16le PCM, encodes EVERY 20 ms frame as the first frame of its own stream in ONE
 * opusgpu_encode_batch() call (BASELINE config #3's definition of an independent frame), decodes the packets again with
 * ONE opusgpu_decode_batch() call to check that encoder and decoder final ranges agree (the reference's own consistency
 * check, tests/test_opus_encode.c:305-306), and writes the packets in opus_demo's container: per frame
 * [be32 length][be32 final range][payload] (src/opus_demo.c:747-765). The HIP runtime is used through its C API for
 * device memory only. Built by concentus_amd/csrc/Makefile with gcc -std=c99. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <hip/hip_runtime_api.h>
#include "../include/opusgpu.h"

static void be32(uint32_t v, unsigned char *p) { p[0] = v >> 24; p[1] = v >> 16; p[2] = v >> 8; p[3] = v; }

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s in.pcm out.bit [bitrate vbr cvbr complexity]\n", argv[0]); return 1; }
    opusgpu_celt_config cfg = {2, 96000, 1, 0, 10, 16, 0, 1500};      /* opus_demo restricted-lowdelay 48000 2 96000 */
    if (argc > 3) cfg.bitrate = atoi(argv[3]);
    if (argc > 4) cfg.vbr = atoi(argv[4]);
    if (argc > 5) cfg.constrained_vbr = atoi(argv[5]);
    if (argc > 6) cfg.complexity = atoi(argv[6]);
    FILE *fi = fopen(argv[1], "rb");
    if (!fi) { perror(argv[1]); return 1; }
    fseek(fi, 0, SEEK_END);
    long bytes = ftell(fi);
    fseek(fi, 0, SEEK_SET);
    const int n = (int)(bytes / (960 * 2 * 2));
    if (n < 1) { fprintf(stderr, "no whole frame in %s\n", argv[1]); return 1; }
    int16_t *pcm = (int16_t *)malloc((size_t)n * 960 * 2 * 2);
    if (fread(pcm, 960 * 2 * 2, (size_t)n, fi) != (size_t)n) { fprintf(stderr, "short read\n"); return 1; }
    fclose(fi);

    const int stride = 1276;
    int16_t *d_pcm, *d_dec;
    unsigned char *d_out;
    int32_t *d_len, *d_ret;
    uint32_t *d_rng, *d_drng;
    void *d_ws, *d_dst;
    const size_t ws = opusgpu_encode_workspace_bytes(n);
    HIP_OK(hipMalloc((void **)&d_pcm, (size_t)n * 3840));
    HIP_OK(hipMalloc((void **)&d_dec, (size_t)n * 3840));
    HIP_OK(hipMalloc((void **)&d_out, (size_t)n * stride));
    HIP_OK(hipMalloc((void **)&d_len, (size_t)n * 4));
    HIP_OK(hipMalloc((void **)&d_ret, (size_t)n * 4));
    HIP_OK(hipMalloc((void **)&d_rng, (size_t)n * 4));
    HIP_OK(hipMalloc((void **)&d_drng, (size_t)n * 4));
    HIP_OK(hipMalloc(&d_ws, ws));
    HIP_OK(hipMalloc(&d_dst, (size_t)n * (size_t)opusgpu_celt_dec_state_size()));
    HIP_OK(hipMemcpy(d_pcm, pcm, (size_t)n * 3840, hipMemcpyHostToDevice));
    HIP_OK(hipMemset(d_out, 0, (size_t)n * stride));

    int rc = opusgpu_encode_batch(&cfg, NULL, d_pcm, d_out, stride, d_len, d_rng, n, d_ws, ws, NULL);
    if (rc != OPUSGPU_OK) { fprintf(stderr, "opusgpu_encode_batch: %s\n", opusgpu_strerror(rc)); return 3; }
    rc = opusgpu_celt_dec_state_init(d_dst, n, NULL);
    if (rc == OPUSGPU_OK) rc = opusgpu_decode_batch(d_dst, d_out, stride, d_len, d_dec, d_ret, d_drng, n, NULL);
    if (rc != OPUSGPU_OK) { fprintf(stderr, "opusgpu_decode_batch: %s\n", opusgpu_strerror(rc)); return 3; }
    HIP_OK(hipDeviceSynchronize());

    unsigned char *out = (unsigned char *)malloc((size_t)n * stride);
    int32_t *len = (int32_t *)malloc((size_t)n * 4), *ret = (int32_t *)malloc((size_t)n * 4);
    uint32_t *rng = (uint32_t *)malloc((size_t)n * 4), *drng = (uint32_t *)malloc((size_t)n * 4);
    HIP_OK(hipMemcpy(out, d_out, (size_t)n * stride, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(len, d_len, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(ret, d_ret, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(rng, d_rng, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(drng, d_drng, (size_t)n * 4, hipMemcpyDeviceToHost));

    FILE *fo = fopen(argv[2], "wb");
    if (!fo) { perror(argv[2]); return 1; }
    long total = 0;
    for (int i = 0; i < n; i++) {
        unsigned char hdr[8];
        if (len[i] < 0) { fprintf(stderr, "frame %d: %s\n", i, opusgpu_strerror(len[i])); return 3; }
        if (ret[i] != 960 || drng[i] != rng[i]) {
            fprintf(stderr, "frame %d: decoder returned %d, final range 0x%08x vs encoder 0x%08x\n", i, ret[i], drng[i], rng[i]);
            return 4;
        }
        be32((uint32_t)len[i], hdr);
        be32(rng[i], hdr + 4);
        fwrite(hdr, 1, 8, fo);
        fwrite(out + (size_t)i * stride, 1, (size_t)len[i], fo);
        total += len[i];
    }
    fclose(fo);
    fprintf(stderr, "%s: %d independent frames, %ld payload bytes (%.1f per frame), encoder/decoder final ranges agree\n",
            opusgpu_get_version_string(), n, total, (double)total / n);
    return 0;
}

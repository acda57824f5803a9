/* opusgpu_diag.h -- entry points of concentus_amd/libopusgpu_diag.so: DIAGNOSTIC builds of the frame kernels with in-kernel
 * stage stamps (s_memtime). Never used for reported throughput, not loaded by the package, not part of the drop-in boundary;
 * tools/stage_profile*.py are the only callers. The library links against libopusgpu.so. */
#ifndef OPUSGPU_DIAG_H
#define OPUSGPU_DIAG_H
#include "opusgpu.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Diagnostic only (never used for reported throughput): the same kernels with in-kernel stage stamps;
 * d_stamps = zero-initialised uint64 [4096][32] cycle totals per stage and workgroup. */
int opusgpu_encode_batch_diag(const opusgpu_celt_config *cfg, const int16_t *d_pcm, unsigned char *d_out,
                              int out_stride, int32_t *d_out_len, uint32_t *d_out_rng, int n_frames,
                              void *d_workspace, size_t workspace_bytes, unsigned long long *d_stamps, void *hip_stream);
/* Diagnostic: stage 1 of opusgpu_decode_batch alone (lane per stream; the states advance, no PCM is produced) with
 * per-stage cycle stamps of each wavefront; d_stamps zero-initialised uint64 [4096][32]. */
int opusgpu_decode_lane_diag(void *d_states, const unsigned char *d_packets, int packet_stride, const int32_t *d_len,
                             int32_t *d_ret, uint32_t *d_rng, int n_streams, unsigned long long *d_stamps, void *hip_stream);
/* Diagnostic: the back phase alone, one lane per frame, with per-stage cycle stamps of each wavefront.
 * Consumes the FrameMid records a preceding opusgpu_encode_batch_diag(same n_frames) left in d_workspace
 * (the records are transformed in place: run it once per encode). d_stamps as above, one row
 * per wavefront of 64 frames. */
int opusgpu_back_lane_diag(const opusgpu_celt_config *cfg, void *d_workspace, unsigned char *d_out, int out_stride,
                           int32_t *d_out_len, uint32_t *d_out_rng, int n_frames, unsigned long long *d_stamps,
                           void *stream);

#ifdef __cplusplus
}
#endif
#endif /* OPUSGPU_DIAG_H */

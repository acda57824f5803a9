// celt_dec_kernel.hip -- batched opus_decode() for CELT-only 20 ms stereo packets, one lane per stream.
//
// Replaces opus_decode() (opus-fix/src/opus_decoder.c:758; include/opus.h:462) -> opus_decode_native ->
// opus_decode_frame -> celt_decode_with_ec (celt/celt_decoder.c:713) for N streams at once. Decoding a
// packet is a serial chain (range decoder -> energies -> allocation -> PVQ), so that part uses the mapping of
// the encoder's back phase: 64 streams share a wavefront (celt_decode_lane_kernel, this file, LANES == 1 build).
// The synthesis has data parallelism inside a frame (denormalisation, inverse MDCT) and runs one wavefront per
// stream (celt_dec_synth_kernel.hip); the post-filter and de-emphasis are recurrences per channel and run one
// lane per (stream, channel) (celt_decode_post_kernel, this file). The stream state (decode_mem, energies,
// post-filter) and the hand-off between the three kernels live in opusgpu_celt_dec_state in HBM.
#define CA_LANE_FRAME 1
#include "celt_lane_tables.h"
#include "celt_dec.h"
#include "opusgpu_internal.h"

namespace ca {

// A = streams per wavefront (64, or 32 / 16 with 2 / 4 partly filled waves per 64-stream workgroup; see
// celt_back_lane_kernel.hip)
template <int A>
__global__ __launch_bounds__(64 * (64 / A)) void celt_decode_lane_kernel(opusgpu_celt_dec_state *states, const u8 *__restrict__ packets,
                                                              int packet_stride, const int *__restrict__ len,
                                                              int *__restrict__ ret, u32 *__restrict__ rng, int n)
{
    fill_lds_tables();
    const int l = threadIdx.x & 63;
    if (l >= A) return;
    const int slot = (threadIdx.x >> 6) * A + l;
    const int k = blockIdx.x * 64 + slot;
    if (k >= n) return;
    DecWork F;
    F.lds_iy16 = (CA_AS_LDS i16 *)(g_lds_iy16 + slot);
    F.lds_pvq16 = (CA_AS_LDS i16 *)(g_lds_pvq16 + slot);
    const int ln = len[k];
    if (ln > packet_stride) {
        // a length past this stream's row would read the next stream's packet (or, for the last stream, past the slab)
        states[k].mid_valid = 0;
        ret[k] = OPUSGPU_BAD_ARG;
        rng[k] = 0;
        return;
    }
    DecResult r = celt_decode_front(F, states + k, packets + (size_t)k * packet_stride, ln);
    ret[k] = r.samples;
    rng[k] = r.final_range;
}

// celt_decode_post_channel (celt_dec.h) for the lane pair (2k, 2k+1) = the two channels of stream k. The post-filter is the
// same code; the de-emphasis (celt_decoder.c:183-285) runs in blocks of 32 samples -- eight 16-byte loads issued together, the
// recurrence over them, then the stores -- and the two lanes exchange their outputs (DPP) so that each writes interleaved
// stereo 16 bytes at a time: a load, the recurrence step and a 2-byte store per sample was one exposed memory round trip per
// sample (loads queue behind stores), 960 per channel and most of this kernel's 0.70 ms.
__device__ __forceinline__ void celt_decode_post_pair(opusgpu_celt_dec_state *st, int c, i16 *pcm)
{
    if (!st->mid_valid) return;
    const int N = FRAME;
    i32 *out_syn = st->decode_mem[c] + DEC_BUF - N;
    comb_filter_inplace_dec(out_syn, st->mid_pf_period_old, st->mid_pf_period, 120, st->mid_pf_gain_old, st->mid_pf_gain,
                            st->mid_pf_tapset_old, st->mid_pf_tapset);
    comb_filter_inplace_dec(out_syn + 120, st->mid_pf_period, st->mid_pf_period_new, N - 120, st->mid_pf_gain, st->mid_pf_gain_new,
                            st->mid_pf_tapset, st->mid_pf_tapset_new);
    i32 m = st->preemph_memD[c];
    const int4 *src = reinterpret_cast<const int4 *>(out_syn);
    for (int j0 = 0; j0 < N; j0 += 32) {
        int4 in[8];
#pragma unroll
        for (int q = 0; q < 8; q++) in[q] = src[(j0 >> 2) + q];
        int4 res[4];
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const i32 w[8] = {in[2 * g].x, in[2 * g].y, in[2 * g].z, in[2 * g].w, in[2 * g + 1].x, in[2 * g + 1].y, in[2 * g + 1].z, in[2 * g + 1].w};
            u32 v[8], o[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const i32 t = add32(w[k], m);
                m = mul16_32_q15(27853, t);
                i32 x = pshr32(t, 12);
                x = imax(x, -32768);
                x = imin(x, 32767);
                v[k] = (u32)x & 0xffffu;
                o[k] = (u32)__builtin_amdgcn_update_dpp(0, (int)v[k], 0xB1, 0xF, 0xF, false);     // quad_perm [1,0,3,2]: the other channel
            }
            // channel 0's lane writes the group's stereo pairs 0..3, channel 1's lane pairs 4..7
            u32 pr[4];
#pragma unroll
            for (int i = 0; i < 4; i++) pr[i] = c ? (o[4 + i] | (v[4 + i] << 16)) : (v[i] | (o[i] << 16));
            res[g] = make_int4((int)pr[0], (int)pr[1], (int)pr[2], (int)pr[3]);
        }
#pragma unroll
        for (int g = 0; g < 4; g++) *reinterpret_cast<int4 *>(pcm + 2 * (j0 + 8 * g + 4 * c)) = res[g];
    }
    st->preemph_memD[c] = m;
}

__global__ __launch_bounds__(256) void celt_decode_post_kernel(opusgpu_celt_dec_state *states, i16 *__restrict__ pcm, int n)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int k = t >> 1, c = t & 1;
    if (k >= n) return;
    celt_decode_post_pair(states + k, c, pcm + (size_t)k * FRAME * 2);
}

// fresh decoder state (opus_decoder_create + OPUS_RESET_STATE, celt_decoder.c:1177-1190)
__global__ void celt_dec_state_init_kernel(opusgpu_celt_dec_state *states, int n)
{
    const int i = blockIdx.x;
    if (i >= n) return;
    u32 *w = reinterpret_cast<u32 *>(&states[i]);
    for (int k = threadIdx.x; k < (int)(sizeof(opusgpu_celt_dec_state) / 4); k += blockDim.x) w[k] = 0;
    __syncthreads();
    for (int k = threadIdx.x; k < 2 * NB; k += blockDim.x) {
        states[i].oldLogE[k] = -28672;
        states[i].oldLogE2[k] = -28672;
    }
}

// quant_all_bands(encode = 0) on its own (bands.c:1337-1502 as called at celt_decoder.c:977), for the per-call hook
// opusgpu_quant_all_bands: ONE lane of the lane build runs quant_all_bands_dec on a record -- the same code the batched
// decoder runs per stream.
__global__ __launch_bounds__(64) void quant_all_bands_dec_hook_kernel(opusgpu_qab_dec_record *rec)
{
    fill_lds_tables();
    if (threadIdx.x != 0) return;
    DecWork F;
    F.lds_iy16 = (CA_AS_LDS i16 *)(g_lds_iy16);
    F.lds_pvq16 = (CA_AS_LDS i16 *)(g_lds_pvq16);
    F.X = (x16_t *)rec->X;
    F.norm = (x16_t *)rec->norm;
    F.diag = nullptr;
    for (int k = 0; k < NB; k++) { F.pulses[k] = rec->pulses[k]; F.tf_res[k] = rec->tf_res[k]; }
    for (int k = 0; k < 2 * NB; k++) F.collapse_masks[k] = 0;
    RangeDec dec;
    dec.buf = rec->buf;
    dec.storage = rec->ec_storage; dec.end_offs = rec->ec_end_offs; dec.end_window = rec->ec_end_window;
    dec.nend_bits = rec->ec_nend_bits; dec.nbits_total = rec->ec_nbits_total; dec.offs = rec->ec_offs; dec.rng = rec->ec_rng;
    dec.val = rec->ec_val; dec.ext = rec->ec_ext; dec.rem = rec->ec_rem; dec.error = rec->ec_error;
    u32 seed = rec->seed;
    quant_all_bands_dec(F, dec, rec->shortBlocks, rec->spread, rec->dual_stereo, rec->intensity, rec->total_bits, rec->balance,
                        rec->codedBands, &seed);
    for (int k = 0; k < 2 * NB; k++) rec->collapse_masks[k] = F.collapse_masks[k];
    rec->seed = seed;
    rec->ec_end_offs = dec.end_offs; rec->ec_end_window = dec.end_window; rec->ec_nend_bits = dec.nend_bits;
    rec->ec_nbits_total = dec.nbits_total; rec->ec_offs = dec.offs; rec->ec_rng = dec.rng; rec->ec_val = dec.val;
    rec->ec_ext = dec.ext; rec->ec_rem = dec.rem; rec->ec_error = dec.error;
}

}  // namespace ca

extern "C" int opusgpu_launch_quant_all_bands_dec(opusgpu_qab_dec_record *d_rec)
{
    hipLaunchKernelGGL(ca::quant_all_bands_dec_hook_kernel, dim3(1), dim3(64), 0, 0, d_rec);
    return opusgpu_check_launch();
}

extern "C" void opusgpu_launch_dec_synth(void *states, int n, hipStream_t s);

extern "C" int opusgpu_celt_dec_state_size(void) { return (int)sizeof(opusgpu_celt_dec_state); }

extern "C" int opusgpu_celt_dec_state_init(void *d_states, int n_streams, void *stream)
{
    if (n_streams < 0 || (n_streams > 0 && !d_states)) return OPUSGPU_BAD_ARG;
    if (n_streams == 0) return OPUSGPU_OK;
    hipLaunchKernelGGL(ca::celt_dec_state_init_kernel, dim3(n_streams), dim3(256), 0, (hipStream_t)stream,
                       (opusgpu_celt_dec_state *)d_states, n_streams);
    return opusgpu_check_launch();
}

extern "C" int opusgpu_decode_batch(void *d_states, const unsigned char *d_packets, int packet_stride, const int32_t *d_len,
                                    int16_t *d_pcm, int32_t *d_ret, uint32_t *d_rng, int n_streams, void *stream)
{
    if (n_streams < 0) return OPUSGPU_BAD_ARG;
    if (n_streams == 0) return OPUSGPU_OK;
    if (!d_states || !d_packets || !d_len || !d_pcm || !d_ret || !d_rng || packet_stride <= 0) return OPUSGPU_BAD_ARG;
    if (((uintptr_t)d_pcm & 15) != 0) return OPUSGPU_BAD_ARG;         // the post kernel stores 16 bytes at a time
    hipStream_t s = (hipStream_t)stream;
    int slot = opusgpu_timing_begin(OPUSGPU_KERNEL_DEC_LANE, s);
    {
        const int a = opusgpu_lane_frames(32);          // (decoder lane kernel: 5.06 ms at 32 streams per wavefront, 5.39 ms at 64)
        const dim3 grid((n_streams + 63) / 64), block(64 * (64 / a));
#define CA_LAUNCH(A) hipLaunchKernelGGL(ca::celt_decode_lane_kernel<A>, grid, block, 0, s, (opusgpu_celt_dec_state *)d_states, \
                                        d_packets, packet_stride, d_len, d_ret, d_rng, n_streams)
        if (a == 32) CA_LAUNCH(32);
        else CA_LAUNCH(64);
#undef CA_LAUNCH
    }
    opusgpu_timing_end(slot, s);
    slot = opusgpu_timing_begin(OPUSGPU_KERNEL_DEC_SYNTH, s);
    opusgpu_launch_dec_synth(d_states, n_streams, s);
    opusgpu_timing_end(slot, s);
    slot = opusgpu_timing_begin(OPUSGPU_KERNEL_DEC_POST, s);
    hipLaunchKernelGGL(ca::celt_decode_post_kernel, dim3((2 * n_streams + 255) / 256), dim3(256), 0, s,
                       (opusgpu_celt_dec_state *)d_states, d_pcm, n_streams);
    opusgpu_timing_end(slot, s);
    return opusgpu_check_launch();
}

// opusgpu_internal.h -- helpers shared by the translation units of libopusgpu.so (not exported API).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/opusgpu.h"

extern "C" void opusgpu_set_last_error(int err);
// hipGetLastError() mapped to an OPUSGPU_* code
extern "C" int opusgpu_check_launch(void);
// optional per-kernel timing (see opusgpu_kernel_timing_enable); slot -1 = timing off
extern "C" int opusgpu_timing_begin(int kernel, hipStream_t s);
extern "C" void opusgpu_timing_end(int slot, hipStream_t s);
// frames (streams) per wavefront of the lane-per-frame kernels: 64 or 32 (OPUSGPU_LANE_FRAMES overrides the kernel's default)
extern "C" int opusgpu_lane_frames(int dflt);
// device counter of SILK records whose header failed the bounds checks (silk_validate.h); nullptr = allocation failed
extern "C" int *opusgpu_bad_record_counter(void);
extern "C" int opusgpu_private_bad_counter_begin(void);
extern "C" int opusgpu_private_bad_counter_end(void);
// Scope of a one-record hook: the record kernels it launches count rejected records into a counter of the calling thread
// (runtime.hip), never into the device's shared one that concurrent batch callers read through opusgpu_silk_bad_records().
struct OpusgpuHookBadScope {
    int rc;
    bool open;
    OpusgpuHookBadScope() : rc(opusgpu_private_bad_counter_begin()), open(true) {}
    int take() { open = false; return opusgpu_private_bad_counter_end(); }
    ~OpusgpuHookBadScope() { if (open) (void)opusgpu_private_bad_counter_end(); }
};
// records per wavefront of the lane-per-record SILK analysis kernels that keep no per-lane LDS: 64, 32, 16 or 8 (OPUSGPU_SILK_LANES)
extern "C" int opusgpu_silk_lanes_per_block(void);
// hipMemcpy of a per-call hook with its status mapped: OPUSGPU_OK or OPUSGPU_INTERNAL_ERROR
static inline int opusgpu_copy(void *dst, const void *src, size_t bytes, hipMemcpyKind kind)
{
    return hipMemcpy(dst, src, bytes, kind) == hipSuccess ? OPUSGPU_OK : OPUSGPU_INTERNAL_ERROR;
}

// quant_all_bands(encode = 0) as a per-call hook: the record the host entry (quant_bands_hook.hip) hands to the decoder's lane
// build (celt_dec_kernel.hip), and the launcher. In: pulses, tf_res, the scalars, seed, the range decoder's fields + the packet
// bytes; out: X (both channels), collapse_masks, seed, the decoder's fields.
struct opusgpu_qab_dec_record {
    int16_t X[2 * 960];
    int16_t norm[2 * 624];             // working storage (the folding source), not an output
    int32_t pulses[21], tf_res[21];
    int32_t shortBlocks, spread, dual_stereo, intensity, total_bits, balance, codedBands;
    uint32_t seed;
    uint32_t ec_storage, ec_end_offs, ec_end_window, ec_offs, ec_rng, ec_val, ec_ext;
    int32_t ec_nend_bits, ec_nbits_total, ec_rem, ec_error, pad;
    unsigned char collapse_masks[2 * 21 + 6];
    unsigned char buf[1280];
};
extern "C" int opusgpu_launch_quant_all_bands_dec(opusgpu_qab_dec_record *d_rec);

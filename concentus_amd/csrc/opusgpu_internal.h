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
// frames (streams) per wavefront of the lane-per-frame kernels: 64, 32 or 16 (OPUSGPU_LANE_FRAMES)
extern "C" int opusgpu_lane_frames(void);
// device counter of SILK records whose header failed the bounds checks (silk_validate.h); nullptr = allocation failed
extern "C" int *opusgpu_bad_record_counter(void);
// records per wavefront of the lane-per-record SILK analysis kernels that keep no per-lane LDS: 64, 32, 16 or 8 (OPUSGPU_SILK_LANES)
extern "C" int opusgpu_silk_lanes_per_block(void);

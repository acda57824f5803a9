// silk_rate_kernels.hip -- the bitrate-control loop of silk_encode_frame_FIX (opus-fix/silk/fixed/encode_frame_FIX.c:263-423) as
// one step per coding pass, one lane per frame; the arithmetic lives in silk_rate_dev.h. The step reads only the frame's
// opusgpu_silk_rate_ctl record and the five range-coder words ec_tell() needs.
#include <string.h>
#include "silk_rate_dev.h"
#include "opusgpu_internal.h"
#include "../../include/opusgpu_silk.h"
#include "silk_validate.h"

namespace ca {

__global__ __launch_bounds__(64) void silk_rate_control_kernel(opusgpu_silk_rate_ctl *__restrict__ ctls, const opusgpu_ec_state *__restrict__ ecs,
                                                               int n_rec, int *__restrict__ bad_records)
{
    const int r = blockIdx.x * 64 + threadIdx.x;
    if (r >= n_rec) return;
    opusgpu_silk_rate_ctl &io = ctls[r];
    if (!rate_ctl_record_ok(io, ecs[r].rng)) {
        io.status = OPUSGPU_BAD_ARG; io.done = 1; io.recode = io.save2 = io.restore2 = 0;
        atomicAdd(bad_records, 1);
        return;
    }
    opusgpu_silk_rate_ctl c = io;                                                            // 160 bytes: registers
    if (c.done) { io.recode = io.save2 = io.restore2 = 0; return; }
    const i32 nBits = ecs[r].nbits_total - (32 - __clz((int)ecs[r].rng));                      // ec_tell (celt/entcode.h:111-113)
    silk_rate_control_step_dev(c, nBits);
    io = c;
}

}  // namespace ca

using namespace ca;

extern "C" int opusgpu_silk_rate_control_batch(opusgpu_silk_rate_ctl *d_ctl, const opusgpu_ec_state *d_ec, int n, void *stream)
{
    if (n < 0) return OPUSGPU_BAD_ARG;
    if (n == 0) return OPUSGPU_OK;
    if (!d_ctl || !d_ec) return OPUSGPU_BAD_ARG;
    int *bad = opusgpu_bad_record_counter();
    if (!bad) return OPUSGPU_ALLOC_FAIL;
    hipLaunchKernelGGL(silk_rate_control_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, d_ctl, d_ec, n, bad);
    return opusgpu_check_launch();
}

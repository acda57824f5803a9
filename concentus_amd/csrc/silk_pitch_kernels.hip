// silk_pitch_kernels.hip -- batched silk_find_pitch_lags_FIX (opus-fix/silk/fixed/find_pitch_lags_FIX.c:37-145) with the three-stage
// pitch estimator, one lane per frame; the arithmetic lives in silk_pitch_dev.h. The whitened buffer goes straight into the
// output record (the estimator reads it back from there).
#include <string.h>
#include "silk_pitch_dev.h"
#include "opusgpu_internal.h"
#include "../../include/opusgpu_silk.h"
#include "silk_validate.h"

namespace ca {

__global__ __launch_bounds__(64) void silk_find_pitch_lags_kernel(const opusgpu_find_pitch_lags_in *__restrict__ recs,
                                                                  opusgpu_find_pitch_lags_out *__restrict__ outs, int n_rec,
                                                                  int *__restrict__ bad_records)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rec) return;
    const opusgpu_find_pitch_lags_in &in = recs[r];
    opusgpu_find_pitch_lags_out &out = outs[r];
    if (!find_pitch_lags_record_ok(in)) {
        memset(&out, 0, sizeof(out));
        out.status = OPUSGPU_BAD_ARG;
        atomicAdd(bad_records, 1);
        return;
    }
    PitchCfg c;
    c.fs_kHz = in.fs_kHz; c.nb_subfr = in.nb_subfr; c.frame_length = in.frame_length; c.ltp_mem_length = in.ltp_mem_length; c.la_pitch = in.la_pitch;
    c.pitch_LPC_win_length = in.pitch_LPC_win_length; c.pitchEstimationLPCOrder = in.pitchEstimationLPCOrder;
    c.pitchEstimationComplexity = in.pitchEstimationComplexity; c.pitchEstimationThreshold_Q16 = in.pitchEstimationThreshold_Q16;
    c.signalType = in.signalType; c.first_frame_after_reset = in.first_frame_after_reset; c.speech_activity_Q8 = in.speech_activity_Q8;
    c.prevSignalType = in.prevSignalType; c.input_tilt_Q15 = in.input_tilt_Q15; c.prevLag = in.prevLag; c.LTPCorr_Q15 = in.LTPCorr_Q15;
    PitchOut o;
    memset(&o, 0, sizeof(o));
    // one work array: the windowed block (384; its down-shifted copy for the autocorrelation is made in place) is dead before the
    // estimator takes its stage-3 copy of the frame (640)
    i16 work[PE_MAX_FRAME];
    const int buf_len = in.la_pitch + in.frame_length + in.ltp_mem_length;
    silk_find_pitch_lags_dev(c, (const i16 *)in.x_buf, (i16 *)out.res, (i16 *)work, (i16 *)work, (i16 *)work, o);
    for (int k = buf_len; k < OPUSGPU_SILK_PITCH_BUF; k++) out.res[k] = 0;
    for (int k = 0; k < 4; k++) out.pitchL[k] = k < in.nb_subfr ? o.pitchL[k] : 0;
    out.lagIndex = o.lagIndex; out.contourIndex = o.contourIndex; out.LTPCorr_Q15 = o.LTPCorr_Q15; out.signalType = o.signalType;
    out.predGain_Q16 = o.predGain_Q16; out.reserved[0] = out.reserved[1] = 0;
    out.status = OPUSGPU_OK;
}

}  // namespace ca

using namespace ca;

extern "C" int opusgpu_silk_find_pitch_lags_batch(const opusgpu_find_pitch_lags_in *d_in, opusgpu_find_pitch_lags_out *d_out, int n, void *stream)
{
    if (n < 0) return OPUSGPU_BAD_ARG;
    if (n == 0) return OPUSGPU_OK;
    if (!d_in || !d_out) return OPUSGPU_BAD_ARG;
    int *bad = opusgpu_bad_record_counter();
    if (!bad) return OPUSGPU_ALLOC_FAIL;
    // Lanes per wavefront: the kernel is bound by the latency of one frame's serial chain (its work arrays live in scratch memory),
    // not by issue slots, so partially filled wavefronts -- more of them per SIMD -- hide that latency (OPUSGPU_SILK_LANES=16/32/64).
    const int lpb = opusgpu_silk_lanes_per_block();
    hipLaunchKernelGGL(silk_find_pitch_lags_kernel, dim3((n + lpb - 1) / lpb), dim3(lpb), 0, (hipStream_t)stream, d_in, d_out, n, bad);
    return opusgpu_check_launch();
}

// celt_dec_synth_kernel.hip -- stage 2 of the batched decoder: celt_synthesis (opus-fix/celt/celt_decoder.c:287-350)
// with one wavefront per stream -- history shift, denormalise_bands (bands.c:169) and the inverse MDCT with TDAC
// (mdct.c:263) straight into the stream's decode_mem in HBM; 7.7 KB of LDS per wavefront.
#include "celt_dec.h"
#include "opusgpu_internal.h"

namespace ca {

__global__ __launch_bounds__(64, 4) void celt_decode_synth_kernel(opusgpu_celt_dec_state *states, int n)
{
    __shared__ SynthLds L;
    for (int k = blockIdx.x; k < n; k += gridDim.x) {
        celt_decode_synth(L, states + k);
        wave_sync();
    }
}

}  // namespace ca

extern "C" void opusgpu_launch_dec_synth(void *states, int n, hipStream_t s)
{
    const int cus = opusgpu_num_cus();
    const int g = n < cus * 16 ? n : cus * 16;
    hipLaunchKernelGGL(ca::celt_decode_synth_kernel, dim3(g), dim3(64), 0, s, (opusgpu_celt_dec_state *)states, n);
}

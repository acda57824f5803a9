// silk_bits_kernels.hip -- batched silk_encode_indices + silk_encode_pulses (opus-fix/silk/encode_indices.c:36, encode_pulses.c:64): side
// information and excitation of one SILK frame onto the Opus range coder, one lane per frame; the arithmetic lives in
// silk_bits_dev.h, the coder in rangecoder.h (lane build: every lane writes its own buffer).
#define CA_LANE_FRAME 1
#include <stdlib.h>
#include <string.h>
#include "fixmath.h"
#include "silk_entropy_tables.h"
// The excitation coder looks its tables up at data-dependent per-lane addresses, once or twice per coded symbol, in one long
// dependent chain: from the constant tables in global memory every look-up is an L1 / L2 round trip of a wavefront that has
// nothing else to do. The workgroup keeps LDS copies and the sources below see those under the tables' names.
#define SILK_BITS_LDS_TABLES(X) \
    X(SILK_shell_code_table0, 152) X(SILK_shell_code_table1, 152) X(SILK_shell_code_table2, 152) X(SILK_shell_code_table3, 152) \
    X(SILK_shell_code_table_offsets, 17) X(SILK_pulses_per_block_iCDF, 180) X(SILK_pulses_per_block_BITS_Q5, 162) \
    X(SILK_rate_levels_iCDF, 18) X(SILK_rate_levels_BITS_Q5, 18) X(SILK_sign_iCDF, 42) X(SILK_lsb_iCDF, 2) X(SILK_max_pulses_table, 4)
namespace ca {
struct SilkBitsTablesLds {
#define X(NAME, N) u8 NAME##_[N];
    SILK_BITS_LDS_TABLES(X)
#undef X
};
__shared__ SilkBitsTablesLds g_bits_tables;
__device__ __forceinline__ void fill_bits_tables()             // followed by the caller's __syncthreads()
{
#define X(NAME, N) for (int k = threadIdx.x; k < N; k += blockDim.x) g_bits_tables.NAME##_[k] = NAME[k];
    SILK_BITS_LDS_TABLES(X)
#undef X
}
}  // namespace ca
#define SILK_shell_code_table0 g_bits_tables.SILK_shell_code_table0_
#define SILK_shell_code_table1 g_bits_tables.SILK_shell_code_table1_
#define SILK_shell_code_table2 g_bits_tables.SILK_shell_code_table2_
#define SILK_shell_code_table3 g_bits_tables.SILK_shell_code_table3_
#define SILK_shell_code_table_offsets g_bits_tables.SILK_shell_code_table_offsets_
#define SILK_pulses_per_block_iCDF g_bits_tables.SILK_pulses_per_block_iCDF_
#define SILK_pulses_per_block_BITS_Q5 g_bits_tables.SILK_pulses_per_block_BITS_Q5_
#define SILK_rate_levels_iCDF g_bits_tables.SILK_rate_levels_iCDF_
#define SILK_rate_levels_BITS_Q5 g_bits_tables.SILK_rate_levels_BITS_Q5_
#define SILK_sign_iCDF g_bits_tables.SILK_sign_iCDF_
#define SILK_lsb_iCDF g_bits_tables.SILK_lsb_iCDF_
#define SILK_max_pulses_table g_bits_tables.SILK_max_pulses_table_
#include "silk_bits_dev.h"
#include "opusgpu_internal.h"
#include "../../include/opusgpu_silk.h"
#include "silk_validate.h"

namespace ca {

__global__ __launch_bounds__(64) void silk_encode_bits_kernel(const opusgpu_silk_bits_in *__restrict__ recs, opusgpu_ec_state *__restrict__ ecs,
                                                              opusgpu_silk_bits_out *__restrict__ outs, int n_rec, int *__restrict__ bad_records,
                                                              const int *__restrict__ rows)
{
    __shared__ NlsfTablesLds tables;
    nlsf_stage_tables(tables, threadIdx.x, blockDim.x);
    fill_bits_tables();
    __syncthreads();
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rec) return;
    if (rows) r = rows[r];                                     // the bitrate loop's second passes: a list of frames, in place
    const opusgpu_silk_bits_in &in = recs[r];
    opusgpu_ec_state &st = ecs[r];
    opusgpu_silk_bits_out o;
    o.ec_prevSignalType = in.ec_prevSignalType; o.ec_prevLagIndex = in.ec_prevLagIndex; o.status = OPUSGPU_OK; o.reserved = 0;
    if (!silk_bits_record_ok(in, st)) {
        o.status = OPUSGPU_BAD_ARG;
        outs[r] = o;
        atomicAdd(bad_records, 1);
        return;
    }
    RangeEnc ec;
    ec.buf = st.buf; ec.storage = st.storage; ec.end_offs = st.end_offs; ec.end_window = st.end_window; ec.nend_bits = st.nend_bits;
    ec.nbits_total = st.nbits_total; ec.offs = st.offs; ec.rng = st.rng; ec.val = st.val; ec.ext = st.ext; ec.rem = st.rem; ec.error = st.error;
    if (in.which & 1) {
        SilkIndices ix;
        for (int k = 0; k < 4; k++) { ix.GainsIndices[k] = in.GainsIndices[k]; ix.LTPIndex[k] = in.LTPIndex[k]; }
        for (int k = 0; k <= SILK_MAX_LPC; k++) ix.NLSFIndices[k] = in.NLSFIndices[k];
        ix.lagIndex = in.lagIndex; ix.contourIndex = in.contourIndex; ix.signalType = in.signalType; ix.quantOffsetType = in.quantOffsetType;
        ix.NLSFInterpCoef_Q2 = in.NLSFInterpCoef_Q2; ix.PERIndex = in.PERIndex; ix.LTP_scaleIndex = in.LTP_scaleIndex; ix.Seed = in.Seed;
        silk_encode_indices_dev(ec, ix, in.nb_subfr, in.fs_kHz, in.predictLPCOrder, in.condCoding, o.ec_prevSignalType, o.ec_prevLagIndex, &tables);
    }
    if (in.which & 2) {
        u8 absq[OPUSGPU_SILK_MAX_FRAME];
        silk_encode_pulses_dev(ec, in.signalType, in.quantOffsetType, (const i8 *)in.pulses, in.frame_length, absq);
    }
    st.end_offs = ec.end_offs; st.end_window = ec.end_window; st.nend_bits = ec.nend_bits; st.nbits_total = ec.nbits_total; st.offs = ec.offs;
    st.rng = ec.rng; st.val = ec.val; st.ext = ec.ext; st.rem = ec.rem; st.error = ec.error;
    outs[r] = o;
}

}  // namespace ca

using namespace ca;

// Records per wavefront: the coder is a chain of dependent loads (the pulses, the tables) with few instructions in between, so
// partially filled wavefronts -- more of them per SIMD -- hide each other's latency (OPUSGPU_SILK_BITS_LANES=16/32/64).
static int bits_lanes()
{
    const char *e = getenv("OPUSGPU_SILK_BITS_LANES");
    const int v = e ? atoi(e) : 64;
    return (v == 16 || v == 32 || v == 64) ? v : 64;
}

extern "C" int opusgpu_silk_encode_bits_batch(const opusgpu_silk_bits_in *d_in, opusgpu_ec_state *d_ec, opusgpu_silk_bits_out *d_out, int n, void *stream)
{
    if (n < 0) return OPUSGPU_BAD_ARG;
    if (n == 0) return OPUSGPU_OK;
    if (!d_in || !d_ec || !d_out) return OPUSGPU_BAD_ARG;
    int *bad = opusgpu_bad_record_counter();
    if (!bad) return OPUSGPU_ALLOC_FAIL;
    const int lpb = bits_lanes();
    hipLaunchKernelGGL(silk_encode_bits_kernel, dim3((n + lpb - 1) / lpb), dim3(lpb), 0, (hipStream_t)stream, d_in, d_ec, d_out, n, bad, (const int *)nullptr);
    return opusgpu_check_launch();
}

// records d_rows[0 .. m) of the arrays, in place: silk_chain.hip's bitrate loop
extern "C" int opusgpu_silk_encode_bits_rows(const opusgpu_silk_bits_in *d_in, opusgpu_ec_state *d_ec, opusgpu_silk_bits_out *d_out, const int *d_rows,
                                             int m, hipStream_t stream)
{
    if (m <= 0) return m < 0 ? OPUSGPU_BAD_ARG : OPUSGPU_OK;
    int *bad = opusgpu_bad_record_counter();
    if (!bad) return OPUSGPU_ALLOC_FAIL;
    const int lpb = bits_lanes();
    hipLaunchKernelGGL(silk_encode_bits_kernel, dim3((m + lpb - 1) / lpb), dim3(lpb), 0, stream, d_in, d_ec, d_out, m, bad, d_rows);
    return opusgpu_check_launch();
}

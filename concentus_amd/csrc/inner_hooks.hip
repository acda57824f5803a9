// inner_hooks.hip -- per-call hooks with the reference's own argument lists (include/opusgpu_hooks.h): opus_ifft,
// comb_filter_const, exp_rotation1, renormalise_vector, silk_NSQ, silk_NSQ_del_dec. Host pointers in and out; each call
// is one small launch between copies (plumbing / parity), the arithmetic is the device code of the batch kernels.
#include <stdlib.h>
#include <string.h>
#include "mdct_dev.h"
#include "celt_math.h"
#include "ec_script.h"
#include "opusgpu_internal.h"
#include "../../include/opusgpu_hooks.h"
#include "../../include/opusgpu_silk.h"

namespace ca {

// ---- opus_ifft_c (kiss_fft.c:602-614) -------------------------------------------------------------------
struct __align__(16) IfftLds {
    MdctLds tab;
    __align__(16) int2 x[480];
};

template <int SHIFT>
__global__ __launch_bounds__(64) void ifft_kernel(const int2 *__restrict__ fin, int2 *__restrict__ fout)
{
    constexpr int NFFT = 480 >> SHIFT;
    __shared__ IfftLds S;
    const int lane = threadIdx.x;
    const MdctTab T = mdct_stage_tables<SHIFT>(S.tab, lane, 64);
    wave_sync();
    for (int i = lane; i < NFFT; i += 64) {
        const int2 v = fin[i];
        fft_put<SHIFT, 1>(S.x, T.bitrev[i], cpx{v.x, neg32(v.y)});
    }
    wave_sync();
    fft_wave<SHIFT, 1>(S.x, T.tw, lane);
    for (int i = lane; i < NFFT; i += 64) {
        const cpx v = fft_get<SHIFT, 1>(S.x, i);
        fout[i] = make_int2(v.r, neg32(v.i));
    }
}

// ---- comb_filter_const_c (celt.c:156-181) -------------------------------------------------------------
// Disjoint y / x: a pure FIR, one output per lane. y == x: the C loop reads samples it has already overwritten once
// i >= T - 2 (the decoder's post-filter), so the loop is run as written, by one lane.
__global__ __launch_bounds__(256) void comb_filter_const_fir_kernel(i32 *__restrict__ y, const i32 *__restrict__ x, int T, int N,
                                                                     i32 g10, i32 g11, i32 g12)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const i32 *p = x + i - T;
    i32 v = add32(x[i], mul16_32_q15(g10, p[0]));
    v = add32(v, mul16_32_q15(g11, add32(p[1], p[-1])));
    v = add32(v, mul16_32_q15(g12, add32(p[2], p[-2])));
    y[i] = v;
}

__global__ void comb_filter_const_inplace_kernel(i32 *x, int T, int N, i32 g10, i32 g11, i32 g12)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    i32 x4 = x[-T - 2], x3 = x[-T - 1], x2 = x[-T], x1 = x[-T + 1];
    for (int i = 0; i < N; i++) {
        const i32 x0 = x[i - T + 2];
        x[i] = add32(add32(add32(x[i], mul16_32_q15(g10, x2)), mul16_32_q15(g11, add32(x1, x3))), mul16_32_q15(g12, add32(x0, x4)));
        x4 = x3; x3 = x2; x2 = x1; x1 = x0;
    }
}

// ---- exp_rotation1 (vq.c:42-68) -----------------------------------------------------------------------
// Positions r, r + stride, ... form independent serial chains (one per residue mod stride): lane r walks chain r, first
// the forward sweep (the value written to X[i + stride] is the x1 of the chain's next step), then the backward sweep.
__global__ __launch_bounds__(64) void exp_rotation1_kernel(i16 *X, int len, int stride, i32 c, i32 s)
{
    __shared__ i16 xs[4096];
    for (int i = threadIdx.x; i < len; i += 64) xs[i] = X[i];
    wave_sync();
    const i32 ms = (i16)neg32(s);
    for (int r = threadIdx.x; r < stride; r += 64) {
        if (r < len - stride) {
            i32 x1 = xs[r];
            int i = r;
            for (; i < len - stride; i += stride) {
                const i32 x2 = xs[i + stride];
                const i32 n2 = (i16)pshr32(mac16_16(mul16_16(c, x2), s, x1), 15);
                xs[i] = (i16)pshr32(mac16_16(mul16_16(c, x1), ms, x2), 15);
                x1 = n2;
            }
            xs[i] = (i16)x1;
        }
    }
    wave_sync();
    const int top = len - 2 * stride - 1;
    for (int r = threadIdx.x; r < stride; r += 64) {
        if (top >= r) {
            int i = top - ((top - r) % stride);
            i32 x2 = xs[i + stride];
            for (; i >= 0; i -= stride) {
                const i32 x1 = xs[i];
                xs[i + stride] = (i16)pshr32(mac16_16(mul16_16(c, x2), s, x1), 15);
                x2 = (i16)pshr32(mac16_16(mul16_16(c, x1), ms, x2), 15);
            }
            xs[i + stride] = (i16)x2;
        }
    }
    wave_sync();
    for (int i = threadIdx.x; i < len; i += 64) X[i] = xs[i];
}

// ---- renormalise_vector (vq.c:347-374) ----------------------------------------------------------------
__global__ __launch_bounds__(64) void renormalise_vector_kernel(i16 *X, int N, i32 gain)
{
    i32 p = 0;
    for (int i = threadIdx.x; i < N; i += 64) p = mac16_16(p, X[i], X[i]);
    const i32 E = add32(1, wave_add(p));                       // EPSILON + celt_inner_prod (wrapping adds commute)
    const int k = celt_ilog2(E) >> 1;
    const i32 t = vshr32(E, 2 * (k - 7));
    const i32 g = (i16)mul16_16_p15(celt_rsqrt_norm(t), gain);
    for (int i = threadIdx.x; i < N; i += 64) X[i] = (i16)pshr32(mul16_16(g, X[i]), k + 1);
}

// ---- range coder scripts (ec_script.h) -------------------------------------------------------------------
struct EcScriptRecord {
    u32 storage, end_offs, end_window, offs, rng, val, ext;
    i32 nend_bits, nbits_total, rem, error, bad_script;
    u8 buf[1280];
};

__global__ __launch_bounds__(64) void ec_enc_script_kernel(EcScriptRecord *rec, const i32 *__restrict__ ops, int n)
{
    __shared__ u8 buf[1280];
    const u32 storage = uni(rec->storage);
    for (u32 k = lane(); k < storage; k += LANES) buf[k] = rec->buf[k];
    RangeEnc e;
    e.buf = buf;
    e.storage = storage; e.end_offs = uni(rec->end_offs); e.end_window = uni(rec->end_window); e.offs = uni(rec->offs);
    e.rng = uni(rec->rng); e.val = uni(rec->val); e.ext = uni(rec->ext); e.nend_bits = uni(rec->nend_bits);
    e.nbits_total = uni(rec->nbits_total); e.rem = uni(rec->rem); e.error = uni(rec->error);
    wave_sync();
    const bool ok = ec_enc_script_ok(ops, n);
    if (ok) ec_enc_run_script(e, ops, n);
    wave_sync();
    for (u32 k = lane(); k < storage; k += LANES) rec->buf[k] = buf[k];
    if (lane() == 0) {
        rec->storage = e.storage; rec->end_offs = e.end_offs; rec->end_window = e.end_window; rec->offs = e.offs; rec->rng = e.rng;
        rec->val = e.val; rec->ext = e.ext; rec->nend_bits = e.nend_bits; rec->nbits_total = e.nbits_total; rec->rem = e.rem;
        rec->error = e.error; rec->bad_script = !ok;
    }
}

__global__ __launch_bounds__(64) void ec_dec_script_kernel(EcScriptRecord *rec, const i32 *__restrict__ ops, int n, i32 *out)
{
    if (threadIdx.x != 0) return;                      // decoding is one serial chain: one lane
    RangeDec d;
    d.buf = rec->buf;
    d.storage = rec->storage; d.end_offs = rec->end_offs; d.end_window = rec->end_window; d.offs = rec->offs; d.rng = rec->rng;
    d.val = rec->val; d.ext = rec->ext; d.nend_bits = rec->nend_bits; d.nbits_total = rec->nbits_total; d.rem = rec->rem;
    d.error = rec->error;
    const bool ok = ec_dec_script_ok(ops, n);
    if (ok) ec_dec_run_script(d, ops, n, out);
    rec->end_offs = d.end_offs; rec->end_window = d.end_window; rec->offs = d.offs; rec->rng = d.rng; rec->val = d.val;
    rec->ext = d.ext; rec->nend_bits = d.nend_bits; rec->nbits_total = d.nbits_total; rec->rem = d.rem; rec->error = d.error;
    rec->bad_script = !ok;
}

}  // namespace ca

using namespace ca;

namespace {
struct DevBuf {
    void *p = nullptr;
    explicit DevBuf(size_t bytes) { if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) p = nullptr; }
    ~DevBuf() { if (p) (void)hipFree(p); }
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
};
bool h2d(void *d, const void *h, size_t n) { return hipMemcpy(d, h, n, hipMemcpyHostToDevice) == hipSuccess; }
bool d2h(void *h, const void *d, size_t n) { return hipMemcpy(h, d, n, hipMemcpyDeviceToHost) == hipSuccess; }
int fail(int rc) { opusgpu_set_last_error(rc); return rc; }
}

struct ref_kiss_fft_state_head2 { int nfft; int16_t scale; int scale_shift; int shift; };

extern "C" void opusgpu_opus_ifft(const void *cfg, const void *fin, void *fout)
{
    const ref_kiss_fft_state_head2 *h = (const ref_kiss_fft_state_head2 *)cfg;
    int shift = -1;
    if (h && fin && fout && fin != fout)
        for (int k = 0; k < 4; k++)
            if (h->nfft == (480 >> k) && h->scale == 17476 && h->scale_shift == 8 - k) shift = k;
    if (shift < 0) { fail(OPUSGPU_BAD_ARG); return; }
    const size_t bytes = (size_t)(480 >> shift) * 8;
    DevBuf din(bytes), dout(bytes);
    if (!din.p || !dout.p) { fail(OPUSGPU_ALLOC_FAIL); return; }
    if (!h2d(din.p, fin, bytes)) { fail(OPUSGPU_INTERNAL_ERROR); return; }
    switch (shift) {
    case 0: hipLaunchKernelGGL(ifft_kernel<0>, dim3(1), dim3(64), 0, 0, (const int2 *)din.p, (int2 *)dout.p); break;
    case 1: hipLaunchKernelGGL(ifft_kernel<1>, dim3(1), dim3(64), 0, 0, (const int2 *)din.p, (int2 *)dout.p); break;
    case 2: hipLaunchKernelGGL(ifft_kernel<2>, dim3(1), dim3(64), 0, 0, (const int2 *)din.p, (int2 *)dout.p); break;
    default: hipLaunchKernelGGL(ifft_kernel<3>, dim3(1), dim3(64), 0, 0, (const int2 *)din.p, (int2 *)dout.p); break;
    }
    int rc = opusgpu_check_launch();
    if (rc == OPUSGPU_OK && !d2h(fout, dout.p, bytes)) rc = OPUSGPU_INTERNAL_ERROR;
    fail(rc);
}

extern "C" void opusgpu_comb_filter_const(int32_t *y, int32_t *x, int T, int N, int g10, int g11, int g12)
{
    if (!y || !x || N < 1 || N > 8192 || T < 3 || T > 4096) { fail(OPUSGPU_BAD_ARG); return; }
    const int back = T + 2;                                      // x[-T-2] is the oldest sample read
    const int32_t *x_lo = x - back, *x_hi = x + N;
    const bool alias = y == x;
    if (!alias && !(y + N <= x_lo || y >= x_hi)) { fail(OPUSGPU_BAD_ARG); return; }      // partial overlap: not meaningful
    const size_t xb = (size_t)(back + N) * 4, yb = (size_t)N * 4;
    DevBuf dx(xb), dy(alias ? 1 : yb);
    if (!dx.p || !dy.p) { fail(OPUSGPU_ALLOC_FAIL); return; }
    if (!h2d(dx.p, x_lo, xb)) { fail(OPUSGPU_INTERNAL_ERROR); return; }
    i32 *d_x = (i32 *)dx.p + back;
    if (alias) hipLaunchKernelGGL(comb_filter_const_inplace_kernel, dim3(1), dim3(64), 0, 0, d_x, T, N, (i32)(i16)g10, (i32)(i16)g11, (i32)(i16)g12);
    else hipLaunchKernelGGL(comb_filter_const_fir_kernel, dim3((N + 255) / 256), dim3(256), 0, 0, (i32 *)dy.p, d_x, T, N,
                            (i32)(i16)g10, (i32)(i16)g11, (i32)(i16)g12);
    int rc = opusgpu_check_launch();
    if (rc == OPUSGPU_OK && !d2h(y, alias ? (void *)d_x : dy.p, yb)) rc = OPUSGPU_INTERNAL_ERROR;
    fail(rc);
}

extern "C" void opusgpu_exp_rotation1(int16_t *X, int len, int stride, int c, int s)
{
    if (!X || len < 2 || len > 4096 || stride < 1 || stride >= len) { fail(OPUSGPU_BAD_ARG); return; }
    DevBuf d((size_t)len * 2);
    if (!d.p) { fail(OPUSGPU_ALLOC_FAIL); return; }
    if (!h2d(d.p, X, (size_t)len * 2)) { fail(OPUSGPU_INTERNAL_ERROR); return; }
    hipLaunchKernelGGL(exp_rotation1_kernel, dim3(1), dim3(64), 0, 0, (i16 *)d.p, len, stride, (i32)(i16)c, (i32)(i16)s);
    int rc = opusgpu_check_launch();
    if (rc == OPUSGPU_OK && !d2h(X, d.p, (size_t)len * 2)) rc = OPUSGPU_INTERNAL_ERROR;
    fail(rc);
}

extern "C" void opusgpu_renormalise_vector(int16_t *X, int N, int gain, int arch)
{
    (void)arch;
    if (!X || N < 1 || N > 4096) { fail(OPUSGPU_BAD_ARG); return; }
    DevBuf d((size_t)N * 2);
    if (!d.p) { fail(OPUSGPU_ALLOC_FAIL); return; }
    if (!h2d(d.p, X, (size_t)N * 2)) { fail(OPUSGPU_INTERNAL_ERROR); return; }
    hipLaunchKernelGGL(renormalise_vector_kernel, dim3(1), dim3(64), 0, 0, (i16 *)d.p, N, (i32)(i16)gain);
    int rc = opusgpu_check_launch();
    if (rc == OPUSGPU_OK && !d2h(X, d.p, (size_t)N * 2)) rc = OPUSGPU_INTERNAL_ERROR;
    fail(rc);
}

// ---- silk_NSQ / silk_NSQ_del_dec: the reference's structs -> one function-boundary record -> the batch kernel -> back --
static int rd_int(const void *base, int off) { int v; memcpy(&v, (const char *)base + off, sizeof(v)); return v; }
static int rd_i8(const void *base, int off) { return (int)*((const int8_t *)base + off); }

static int fill_nsq_record(opusgpu_nsq_in *r, const void *psEncC, const void *psIndices, const int32_t x_Q3[],
                           const int16_t PredCoef_Q12[], const int16_t LTPCoef_Q14[], const int16_t AR2_Q13[],
                           const int HarmShapeGain_Q14[], const int Tilt_Q14[], const int32_t LF_shp_Q14[],
                           const int32_t Gains_Q16[], const int pitchL[], int Lambda_Q10, int LTP_scale_Q14)
{
    memset(r, 0, sizeof(*r));
    r->nb_subfr = rd_int(psEncC, OPUSGPU_REF_OFF_NB_SUBFR);
    r->subfr_length = rd_int(psEncC, OPUSGPU_REF_OFF_SUBFR_LENGTH);
    r->frame_length = rd_int(psEncC, OPUSGPU_REF_OFF_FRAME_LENGTH);
    r->ltp_mem_length = rd_int(psEncC, OPUSGPU_REF_OFF_LTP_MEM_LENGTH);
    r->predictLPCOrder = rd_int(psEncC, OPUSGPU_REF_OFF_PREDICT_LPC_ORDER);
    r->shapingLPCOrder = rd_int(psEncC, OPUSGPU_REF_OFF_SHAPING_LPC_ORDER);
    r->signalType = rd_i8(psIndices, OPUSGPU_REF_OFF_SIGNAL_TYPE);
    r->quantOffsetType = rd_i8(psIndices, OPUSGPU_REF_OFF_QUANT_OFFSET_TYPE);
    r->NLSFInterpCoef_Q2 = rd_i8(psIndices, OPUSGPU_REF_OFF_NLSF_INTERP_COEF_Q2);
    r->Seed = rd_i8(psIndices, OPUSGPU_REF_OFF_SEED);
    if (r->nb_subfr < 1 || r->nb_subfr > 4 || r->frame_length < 1 || r->frame_length > OPUSGPU_SILK_MAX_FRAME) return OPUSGPU_BAD_ARG;
    r->Lambda_Q10 = Lambda_Q10;
    r->LTP_scale_Q14 = LTP_scale_Q14;
    for (int k = 0; k < r->nb_subfr; k++) {
        r->HarmShapeGain_Q14[k] = HarmShapeGain_Q14[k]; r->Tilt_Q14[k] = Tilt_Q14[k]; r->LF_shp_Q14[k] = LF_shp_Q14[k];
        r->Gains_Q16[k] = Gains_Q16[k]; r->pitchL[k] = pitchL[k];
    }
    memcpy(r->x_Q3, x_Q3, sizeof(int32_t) * (size_t)r->frame_length);
    memcpy(r->PredCoef_Q12, PredCoef_Q12, sizeof(r->PredCoef_Q12));
    memcpy(r->LTPCoef_Q14, LTPCoef_Q14, sizeof(int16_t) * 5 * (size_t)r->nb_subfr);
    memcpy(r->AR2_Q13, AR2_Q13, sizeof(int16_t) * 16 * (size_t)r->nb_subfr);
    return OPUSGPU_OK;
}

static void nsq_hook(int del_dec, const void *psEncC, void *NSQ, void *psIndices, const int32_t x_Q3[], int8_t pulses[],
                     const int16_t PredCoef_Q12[], const int16_t LTPCoef_Q14[], const int16_t AR2_Q13[],
                     const int HarmShapeGain_Q14[], const int Tilt_Q14[], const int32_t LF_shp_Q14[],
                     const int32_t Gains_Q16[], const int pitchL[], int Lambda_Q10, int LTP_scale_Q14)
{
    if (!psEncC || !NSQ || !psIndices || !x_Q3 || !pulses || !PredCoef_Q12 || !LTPCoef_Q14 || !AR2_Q13 || !HarmShapeGain_Q14 ||
        !Tilt_Q14 || !LF_shp_Q14 || !Gains_Q16 || !pitchL) { fail(OPUSGPU_BAD_ARG); return; }
    opusgpu_nsq_dd_in rec;
    memset(&rec, 0, sizeof(rec));
    int rc = fill_nsq_record(&rec.base, psEncC, psIndices, x_Q3, PredCoef_Q12, LTPCoef_Q14, AR2_Q13, HarmShapeGain_Q14, Tilt_Q14,
                             LF_shp_Q14, Gains_Q16, pitchL, Lambda_Q10, LTP_scale_Q14);
    if (rc != OPUSGPU_OK) { fail(rc); return; }
    rec.nStatesDelayedDecision = rd_int(psEncC, OPUSGPU_REF_OFF_N_STATES_DEL_DEC);
    rec.warping_Q16 = rd_int(psEncC, OPUSGPU_REF_OFF_WARPING_Q16);
    const size_t in_bytes = del_dec ? sizeof(opusgpu_nsq_dd_in) : sizeof(opusgpu_nsq_in);
    const size_t out_bytes = del_dec ? sizeof(opusgpu_nsq_dd_out) : sizeof(opusgpu_nsq_out);
    const size_t ws_bytes = del_dec ? opusgpu_silk_nsq_del_dec_workspace_bytes(1) : opusgpu_silk_nsq_workspace_bytes(1);
    DevBuf din(in_bytes), dst(sizeof(opusgpu_nsq_state)), dout(out_bytes), dws(ws_bytes);
    if (!din.p || !dst.p || !dout.p || !dws.p) { fail(OPUSGPU_ALLOC_FAIL); return; }
    if (!h2d(din.p, &rec, in_bytes) || !h2d(dst.p, NSQ, sizeof(opusgpu_nsq_state))) { fail(OPUSGPU_INTERNAL_ERROR); return; }
    OpusgpuHookBadScope bad;                 // the out records carry no status word: count rejections privately to this thread
    if (bad.rc != OPUSGPU_OK) { fail(bad.rc); return; }
    rc = del_dec ? opusgpu_silk_nsq_del_dec_batch((const opusgpu_nsq_dd_in *)din.p, (opusgpu_nsq_state *)dst.p, (opusgpu_nsq_dd_out *)dout.p,
                                                  1, dws.p, ws_bytes, nullptr)
                 : opusgpu_silk_nsq_batch((const opusgpu_nsq_in *)din.p, (opusgpu_nsq_state *)dst.p, (opusgpu_nsq_out *)dout.p, 1, dws.p,
                                          ws_bytes, nullptr);
    const int n_bad = bad.take();
    if (rc == OPUSGPU_OK && n_bad != 0) rc = n_bad < 0 ? n_bad : OPUSGPU_BAD_ARG;      // header outside the kernels' bounds
    opusgpu_nsq_dd_out h_out;
    if (rc == OPUSGPU_OK && (!d2h(&h_out, dout.p, out_bytes) || !d2h(NSQ, dst.p, sizeof(opusgpu_nsq_state)))) rc = OPUSGPU_INTERNAL_ERROR;
    fail(rc);
    if (rc != OPUSGPU_OK) return;
    memcpy(pulses, h_out.pulses, (size_t)rec.base.frame_length);
    if (del_dec) *((int8_t *)psIndices + OPUSGPU_REF_OFF_SEED) = (int8_t)h_out.Seed;
}

extern "C" void opusgpu_silk_NSQ(const void *psEncC, void *NSQ, void *psIndices, const int32_t x_Q3[], int8_t pulses[],
                                 const int16_t PredCoef_Q12[], const int16_t LTPCoef_Q14[], const int16_t AR2_Q13[],
                                 const int HarmShapeGain_Q14[], const int Tilt_Q14[], const int32_t LF_shp_Q14[],
                                 const int32_t Gains_Q16[], const int pitchL[], const int Lambda_Q10, const int LTP_scale_Q14)
{
    nsq_hook(0, psEncC, NSQ, psIndices, x_Q3, pulses, PredCoef_Q12, LTPCoef_Q14, AR2_Q13, HarmShapeGain_Q14, Tilt_Q14, LF_shp_Q14,
             Gains_Q16, pitchL, Lambda_Q10, LTP_scale_Q14);
}

extern "C" void opusgpu_silk_NSQ_del_dec(const void *psEncC, void *NSQ, void *psIndices, const int32_t x_Q3[], int8_t pulses[],
                                         const int16_t PredCoef_Q12[], const int16_t LTPCoef_Q14[], const int16_t AR2_Q13[],
                                         const int HarmShapeGain_Q14[], const int Tilt_Q14[], const int32_t LF_shp_Q14[],
                                         const int32_t Gains_Q16[], const int pitchL[], const int Lambda_Q10, const int LTP_scale_Q14)
{
    nsq_hook(1, psEncC, NSQ, psIndices, x_Q3, pulses, PredCoef_Q12, LTPCoef_Q14, AR2_Q13, HarmShapeGain_Q14, Tilt_Q14, LF_shp_Q14,
             Gains_Q16, pitchL, Lambda_Q10, LTP_scale_Q14);
}

// ---- range coder scripts -------------------------------------------------------------------------------------
// the tree's ec_ctx, x86-64 (celt/entcode.h:63-94, with the trailing EC_DIFF of this tree)
namespace {
struct ref_ec_ctx_s {
    unsigned char *buf;
    uint32_t storage, end_offs, end_window;
    int nend_bits, nbits_total;
    uint32_t offs, rng, val, ext;
    int rem, error, EC_DIFF;
};

int run_ec_script(void *ec, const int32_t *ops, int n, int32_t *out, bool decode)
{
    ref_ec_ctx_s *e = (ref_ec_ctx_s *)ec;
    if (!e || !e->buf || n < 0 || (n > 0 && !ops) || (decode && n > 0 && !out)) return fail(OPUSGPU_BAD_ARG);
    if (e->storage > sizeof(((EcScriptRecord *)nullptr)->buf) || n > 65536) return fail(OPUSGPU_UNIMPLEMENTED);
    if (n == 0) return fail(OPUSGPU_OK);
    EcScriptRecord *h = (EcScriptRecord *)calloc(1, sizeof(EcScriptRecord));
    if (!h) return fail(OPUSGPU_ALLOC_FAIL);
    h->storage = e->storage; h->end_offs = e->end_offs; h->end_window = e->end_window; h->offs = e->offs; h->rng = e->rng; h->val = e->val;
    h->ext = e->ext; h->nend_bits = e->nend_bits; h->nbits_total = e->nbits_total; h->rem = e->rem; h->error = e->error;
    memcpy(h->buf, e->buf, e->storage);
    DevBuf drec(sizeof(*h)), dops((size_t)n * 16), dout((size_t)n * 4);
    int rc = (drec.p && dops.p && dout.p) ? OPUSGPU_OK : OPUSGPU_ALLOC_FAIL;
    if (rc == OPUSGPU_OK && !(h2d(drec.p, h, sizeof(*h)) && h2d(dops.p, ops, (size_t)n * 16))) rc = OPUSGPU_INTERNAL_ERROR;
    if (rc == OPUSGPU_OK) {
        if (decode) hipLaunchKernelGGL(ca::ec_dec_script_kernel, dim3(1), dim3(64), 0, 0, (EcScriptRecord *)drec.p, (const i32 *)dops.p, n, (i32 *)dout.p);
        else hipLaunchKernelGGL(ca::ec_enc_script_kernel, dim3(1), dim3(64), 0, 0, (EcScriptRecord *)drec.p, (const i32 *)dops.p, n);
        rc = opusgpu_check_launch();
    }
    if (rc == OPUSGPU_OK && !d2h(h, drec.p, sizeof(*h))) rc = OPUSGPU_INTERNAL_ERROR;
    if (rc == OPUSGPU_OK && h->bad_script) rc = OPUSGPU_BAD_ARG;           // an argument the reference's celt_assert()s reject: nothing ran
    if (rc == OPUSGPU_OK && decode && !d2h(out, dout.p, (size_t)n * 4)) rc = OPUSGPU_INTERNAL_ERROR;
    if (rc == OPUSGPU_OK) {
        if (!decode) memcpy(e->buf, h->buf, e->storage);
        e->storage = h->storage; e->end_offs = h->end_offs; e->end_window = h->end_window; e->offs = h->offs; e->rng = h->rng; e->val = h->val;
        e->ext = h->ext; e->nend_bits = h->nend_bits; e->nbits_total = h->nbits_total; e->rem = h->rem; e->error = h->error;
    }
    free(h);
    return fail(rc);
}
}  // namespace

extern "C" int opusgpu_ec_enc_script(void *ec, const int32_t *ops, int n_ops) { return run_ec_script(ec, ops, n_ops, nullptr, false); }
extern "C" int opusgpu_ec_dec_script(void *ec, const int32_t *ops, int n_ops, int32_t *out) { return run_ec_script(ec, ops, n_ops, out, true); }

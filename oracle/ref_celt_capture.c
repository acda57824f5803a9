/* oracle/ref_celt_capture.c -- TEST INFRASTRUCTURE ONLY.
 * Linked with -Wl,--wrap=quant_all_bands into a capture variant of the compiled reference
 * (oracle/_ref/libopus_ref_celtcap.so): every call of quant_all_bands (opus-fix/celt/bands.c:1337) made by the unmodified
 * reference encoder is forwarded to the real code, and its arguments (copied BEFORE the call: X and Y are transformed in
 * place) and the range coder's state and buffer AFTER the call are recorded, so that the per-call hook
 * opusgpu_quant_all_bands can be checked against the reference's own function on the reference's own inputs
 * (tests/test_hooks_gpu.py). The calls the reference DEcoder makes (encode == 0, celt_decoder.c:977) are recorded likewise: inputs
 * before, decoded bands / collapse masks / seed / range decoder after. Compiled against the reference's headers where they lie; nothing is copied. */
#include <stdlib.h>
#include <string.h>
#include "bands.h"

typedef struct {
    opus_int16 X[960], Y[960];
    opus_int32 bandE[42];
    int pulses[21], tf_res[21];
    int encode, start, end, stereo, shortBlocks, spread, dual_stereo, intensity, LM, codedBands;
    opus_int32 total_bits, balance;
    opus_uint32 seed;
    /* ec_ctx without the buffer pointer, before and after: storage end_offs end_window nend_bits nbits_total offs rng val ext rem error */
    opus_int32 ec_in[11], ec_out[11];
    unsigned char buf_in[1280], buf_out[1280];
} refcap_qab;

static refcap_qab *g_q; static int g_nq, g_capq;
void refcap_start_qab(int max_records) { g_capq = max_records; g_nq = 0; g_q = (refcap_qab *)calloc(max_records, sizeof(*g_q)); }
int refcap_count_qab(void) { return g_nq; }
int refcap_sizeof_qab(void) { return (int)sizeof(refcap_qab); }
void refcap_get_qab(void *dst) { memcpy(dst, g_q, (size_t)g_nq * sizeof(*g_q)); }

/* decoder side (celt_decoder.c:977, encode == 0): inputs before the call, and what the call produces -- the decoded normalised
 * bands, the collapse masks, the LCG seed and the range decoder's state */
typedef struct {
    int pulses[21], tf_res[21];
    int shortBlocks, spread, dual_stereo, intensity, LM, codedBands;
    opus_int32 total_bits, balance;
    opus_uint32 seed_in, seed_out;
    opus_int32 ec_in[11], ec_out[11];
    unsigned char buf[1280];
    opus_int16 X_out[960], Y_out[960];
    unsigned char collapse_masks_out[42], pad[2];
} refcap_qabd;

static refcap_qabd *g_d; static int g_nd, g_capd;
void refcap_start_qab_dec(int max_records) { g_capd = max_records; g_nd = 0; g_d = (refcap_qabd *)calloc(max_records, sizeof(*g_d)); }
int refcap_count_qab_dec(void) { return g_nd; }
int refcap_sizeof_qab_dec(void) { return (int)sizeof(refcap_qabd); }
void refcap_get_qab_dec(void *dst) { memcpy(dst, g_d, (size_t)g_nd * sizeof(*g_d)); }

static void ec_pack(opus_int32 *d, const ec_ctx *e)
{
    d[0] = e->storage; d[1] = e->end_offs; d[2] = e->end_window; d[3] = e->nend_bits; d[4] = e->nbits_total; d[5] = e->offs;
    d[6] = e->rng; d[7] = e->val; d[8] = e->ext; d[9] = e->rem; d[10] = e->error;
}

void __real_quant_all_bands(int encode, const CELTMode *m, int start, int end, celt_norm *X, celt_norm *Y, unsigned char *collapse_masks,
                            const celt_ener *bandE, int *pulses, int shortBlocks, int spread, int dual_stereo, int intensity, int *tf_res,
                            opus_int32 total_bits, opus_int32 balance, ec_ctx *ec, int LM, int codedBands, opus_uint32 *seed, int arch);
void __wrap_quant_all_bands(int encode, const CELTMode *m, int start, int end, celt_norm *X, celt_norm *Y, unsigned char *collapse_masks,
                            const celt_ener *bandE, int *pulses, int shortBlocks, int spread, int dual_stereo, int intensity, int *tf_res,
                            opus_int32 total_bits, opus_int32 balance, ec_ctx *ec, int LM, int codedBands, opus_uint32 *seed, int arch)
{
    refcap_qab *r = (g_q && g_nq < g_capq && encode && Y && LM == 3 && m->nbEBands == 21 && ec->storage <= 1275) ? &g_q[g_nq] : NULL;
    if (r) {
        memcpy(r->X, X, sizeof(r->X)); memcpy(r->Y, Y, sizeof(r->Y)); memcpy(r->bandE, bandE, sizeof(r->bandE));
        memcpy(r->pulses, pulses, sizeof(r->pulses)); memcpy(r->tf_res, tf_res, sizeof(r->tf_res));
        r->encode = encode; r->start = start; r->end = end; r->stereo = 1; r->shortBlocks = shortBlocks; r->spread = spread;
        r->dual_stereo = dual_stereo; r->intensity = intensity; r->LM = LM; r->codedBands = codedBands;
        r->total_bits = total_bits; r->balance = balance; r->seed = *seed;
        ec_pack(r->ec_in, ec);
        memcpy(r->buf_in, ec->buf, ec->storage);
    }
    refcap_qabd *d = (g_d && g_nd < g_capd && !encode && Y && LM == 3 && start == 0 && end == 21 && m->nbEBands == 21 && ec->storage <= 1275) ? &g_d[g_nd] : NULL;
    if (d) {
        memcpy(d->pulses, pulses, sizeof(d->pulses)); memcpy(d->tf_res, tf_res, sizeof(d->tf_res));
        d->shortBlocks = shortBlocks; d->spread = spread; d->dual_stereo = dual_stereo; d->intensity = intensity; d->LM = LM;
        d->codedBands = codedBands; d->total_bits = total_bits; d->balance = balance; d->seed_in = *seed;
        ec_pack(d->ec_in, ec);
        memcpy(d->buf, ec->buf, ec->storage);
    }
    __real_quant_all_bands(encode, m, start, end, X, Y, collapse_masks, bandE, pulses, shortBlocks, spread, dual_stereo, intensity, tf_res,
                           total_bits, balance, ec, LM, codedBands, seed, arch);
    if (d) {
        ec_pack(d->ec_out, ec);
        d->seed_out = *seed;
        memcpy(d->X_out, X, sizeof(d->X_out)); memcpy(d->Y_out, Y, sizeof(d->Y_out));
        memcpy(d->collapse_masks_out, collapse_masks, 42);
        g_nd++;
    }
    if (r) {
        ec_pack(r->ec_out, ec);
        memcpy(r->buf_out, ec->buf, ec->storage);
        g_nq++;
    }
}

// quant_bands_hook.hip -- quant_all_bands() with the reference's own argument list as a per-call hook (include/opusgpu_hooks.h).
//
// One wavefront, the wave-per-frame build of the encoder sources: the caller's arguments are packed into one record, the
// kernel unpacks it into the BackLds working set (normalised bands, band energies, pulses, tf_res, the range coder's buffer),
// runs quant_all_bands_wave (celt_enc_back.h: bands.c:1337-1502 with quant_band / quant_band_stereo / quant_partition /
// compute_theta / alg_quant / encode_pulses and the ec_enc_* calls they make) and hands the coder state and buffer back.
#include <stdlib.h>
#include <string.h>
#include "celt_enc.h"
#include "opusgpu_internal.h"
#include "../../include/opusgpu_hooks.h"

namespace ca {

struct QabRecord {
    i16 X[2 * FRAME];
    i32 bandE[2 * NB];
    i32 pulses[NB], tf_res[NB];
    i32 shortBlocks, spread, dual_stereo, intensity, total_bits, balance, codedBands, pad;
    u32 ec_storage, ec_end_offs, ec_end_window, ec_offs, ec_rng, ec_val, ec_ext;
    i32 ec_nend_bits, ec_nbits_total, ec_rem, ec_error, pad2;
    u8 buf[1280];
};

__global__ __launch_bounds__(64) void quant_all_bands_hook_kernel(QabRecord *rec)
{
    __shared__ BackLds F;
    if (lane() == 0) F.diag = nullptr;
    for (int k = lane(); k < 2 * FRAME; k += LANES) F.x16[k] = rec->X[k];
    for (int k = lane(); k < 2 * NB; k += LANES) F.bandE[k] = rec->bandE[k];
    for (int k = lane(); k < NB; k += LANES) { F.pulses[k] = rec->pulses[k]; F.tf_res[k] = rec->tf_res[k]; }
    const u32 storage = uni(rec->ec_storage);
    for (u32 k = lane(); k < storage; k += LANES) F.packet[1 + k] = rec->buf[k];
    RangeEnc enc;
    enc.buf = F.packet + 1;
    enc.storage = storage; enc.end_offs = uni(rec->ec_end_offs); enc.end_window = uni(rec->ec_end_window);
    enc.offs = uni(rec->ec_offs); enc.rng = uni(rec->ec_rng); enc.val = uni(rec->ec_val); enc.ext = uni(rec->ec_ext);
    enc.nend_bits = uni(rec->ec_nend_bits); enc.nbits_total = uni(rec->ec_nbits_total); enc.rem = uni(rec->ec_rem);
    enc.error = uni(rec->ec_error);
    wave_sync();
    quant_all_bands_wave(F, enc, 2, uni(rec->shortBlocks), uni(rec->spread), uni(rec->dual_stereo), uni(rec->intensity),
                         uni(rec->total_bits), uni(rec->balance), uni(rec->codedBands));
    wave_sync();
    for (u32 k = lane(); k < storage; k += LANES) rec->buf[k] = F.packet[1 + k];
    if (lane() == 0) {
        rec->ec_end_offs = enc.end_offs; rec->ec_end_window = enc.end_window; rec->ec_offs = enc.offs; rec->ec_rng = enc.rng;
        rec->ec_val = enc.val; rec->ec_ext = enc.ext; rec->ec_nend_bits = enc.nend_bits; rec->ec_nbits_total = enc.nbits_total;
        rec->ec_rem = enc.rem; rec->ec_error = enc.error;
    }
}

}  // namespace ca

// the tree's ec_ctx, x86-64 (celt/entcode.h:63-94, with the trailing EC_DIFF of this tree)
struct ref_ec_ctx {
    unsigned char *buf;
    uint32_t storage, end_offs, end_window;
    int nend_bits, nbits_total;
    uint32_t offs, rng, val, ext;
    int rem, error, EC_DIFF;
};
// head of CELTMode (celt/modes.h:52-60)
struct ref_celt_mode_head { int32_t Fs; int overlap; int nbEBands; int effEBands; };

extern "C" void opusgpu_quant_all_bands(int encode, const void *m, int start, int end, int16_t *X, int16_t *Y, unsigned char *collapse_masks,
                                        const int32_t *bandE, int *pulses, int shortBlocks, int spread, int dual_stereo, int intensity,
                                        int *tf_res, int32_t total_bits, int32_t balance, void *ec, int LM, int codedBands, uint32_t *seed,
                                        int arch)
{
    (void)arch;
    if (!m || !X || (encode && !bandE) || !pulses || !tf_res || !ec) { opusgpu_set_last_error(OPUSGPU_BAD_ARG); return; }
    const ref_celt_mode_head *mh = (const ref_celt_mode_head *)m;
    ref_ec_ctx *e = (ref_ec_ctx *)ec;
    if ((encode != 0 && encode != 1) || !Y || start != 0 || end != 21 || LM != 3 || mh->Fs != 48000 || mh->overlap != 120 || mh->nbEBands != 21 ||
        e->storage > 1275 || !e->buf || (shortBlocks != 0 && shortBlocks != 8)) {
        opusgpu_set_last_error(OPUSGPU_UNIMPLEMENTED);
        return;
    }
    if (!encode) {
        // decoder side (celt_decoder.c:977): the range DEcoder reads ec->buf, X / Y receive the decoded normalised bands,
        // collapse_masks and *seed are outputs the caller uses afterwards (anti_collapse, st->rng)
        if (!collapse_masks || !seed) { opusgpu_set_last_error(OPUSGPU_BAD_ARG); return; }
        opusgpu_qab_dec_record *hp = (opusgpu_qab_dec_record *)calloc(1, sizeof(opusgpu_qab_dec_record)), *d = nullptr;
        if (!hp) { opusgpu_set_last_error(OPUSGPU_ALLOC_FAIL); return; }
        for (int k = 0; k < 21; k++) { hp->pulses[k] = pulses[k]; hp->tf_res[k] = tf_res[k]; }
        hp->shortBlocks = shortBlocks; hp->spread = spread; hp->dual_stereo = dual_stereo; hp->intensity = intensity;
        hp->total_bits = total_bits; hp->balance = balance; hp->codedBands = codedBands; hp->seed = *seed;
        hp->ec_storage = e->storage; hp->ec_end_offs = e->end_offs; hp->ec_end_window = e->end_window; hp->ec_offs = e->offs;
        hp->ec_rng = e->rng; hp->ec_val = e->val; hp->ec_ext = e->ext; hp->ec_nend_bits = e->nend_bits;
        hp->ec_nbits_total = e->nbits_total; hp->ec_rem = e->rem; hp->ec_error = e->error;
        memcpy(hp->buf, e->buf, e->storage);
        int rc = hipMalloc(&d, sizeof(*hp)) == hipSuccess ? OPUSGPU_OK : OPUSGPU_ALLOC_FAIL;
        if (rc == OPUSGPU_OK) rc = opusgpu_copy(d, hp, sizeof(*hp), hipMemcpyHostToDevice);
        if (rc == OPUSGPU_OK) rc = opusgpu_launch_quant_all_bands_dec(d);
        if (rc == OPUSGPU_OK) rc = opusgpu_copy(hp, d, sizeof(*hp), hipMemcpyDeviceToHost);
        if (d) (void)hipFree(d);
        opusgpu_set_last_error(rc);
        if (rc == OPUSGPU_OK) {
            memcpy(X, hp->X, sizeof(int16_t) * 960);
            memcpy(Y, hp->X + 960, sizeof(int16_t) * 960);
            memcpy(collapse_masks, hp->collapse_masks, 42);
            *seed = hp->seed;
            e->end_offs = hp->ec_end_offs; e->end_window = hp->ec_end_window; e->offs = hp->ec_offs; e->rng = hp->ec_rng;
            e->val = hp->ec_val; e->ext = hp->ec_ext; e->nend_bits = hp->ec_nend_bits; e->nbits_total = hp->ec_nbits_total;
            e->rem = hp->ec_rem; e->error = hp->ec_error;
        }
        free(hp);
        return;
    }
    (void)collapse_masks; (void)seed;
    static_assert(sizeof(((ca::QabRecord *)nullptr)->buf) >= 1276, "range coder buffer");
    ca::QabRecord *hp = (ca::QabRecord *)calloc(1, sizeof(ca::QabRecord));      // 6.4 KB: kept off a codec thread's stack
    if (!hp) { opusgpu_set_last_error(OPUSGPU_ALLOC_FAIL); return; }
    ca::QabRecord &h = *hp;
    memcpy(h.X, X, sizeof(int16_t) * 960);
    memcpy(h.X + 960, Y, sizeof(int16_t) * 960);
    memcpy(h.bandE, bandE, sizeof(int32_t) * 42);
    for (int k = 0; k < 21; k++) { h.pulses[k] = pulses[k]; h.tf_res[k] = tf_res[k]; }
    h.shortBlocks = shortBlocks; h.spread = spread; h.dual_stereo = dual_stereo; h.intensity = intensity;
    h.total_bits = total_bits; h.balance = balance; h.codedBands = codedBands;
    h.ec_storage = e->storage; h.ec_end_offs = e->end_offs; h.ec_end_window = e->end_window; h.ec_offs = e->offs; h.ec_rng = e->rng;
    h.ec_val = e->val; h.ec_ext = e->ext; h.ec_nend_bits = e->nend_bits; h.ec_nbits_total = e->nbits_total; h.ec_rem = e->rem;
    h.ec_error = e->error;
    memcpy(h.buf, e->buf, e->storage);
    ca::QabRecord *d = nullptr;
    if (hipMalloc(&d, sizeof(h)) != hipSuccess) { free(hp); opusgpu_set_last_error(OPUSGPU_ALLOC_FAIL); return; }
    int rc = hipMemcpy(d, &h, sizeof(h), hipMemcpyHostToDevice) == hipSuccess ? OPUSGPU_OK : OPUSGPU_INTERNAL_ERROR;
    if (rc == OPUSGPU_OK) {
        hipLaunchKernelGGL(ca::quant_all_bands_hook_kernel, dim3(1), dim3(64), 0, 0, d);
        rc = opusgpu_check_launch();
    }
    if (rc == OPUSGPU_OK && hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) rc = OPUSGPU_INTERNAL_ERROR;
    (void)hipFree(d);
    opusgpu_set_last_error(rc);
    if (rc != OPUSGPU_OK) { free(hp); return; }
    memcpy(e->buf, h.buf, e->storage);
    e->end_offs = h.ec_end_offs; e->end_window = h.ec_end_window; e->offs = h.ec_offs; e->rng = h.ec_rng; e->val = h.ec_val;
    e->ext = h.ec_ext; e->nend_bits = h.ec_nend_bits; e->nbits_total = h.ec_nbits_total; e->rem = h.ec_rem; e->error = h.ec_error;
    free(hp);
}

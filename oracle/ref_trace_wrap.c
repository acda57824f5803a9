/* oracle/ref_trace_wrap.c -- TEST/DEBUG INFRASTRUCTURE ONLY.
 * Linked with -Wl,--wrap=ec_enc_init into a *trace variant* of the compiled reference
 * (oracle/_ref/libopus_ref_trace.so): switches on the reference's own EC_DIFF range-coder trace
 * (opus-fix/celt/entcode.h:92-93, printed from celt/entenc.c) for every encoder it initialises, so a
 * symbol-by-symbol log of the reference can be diffed against the kernel's. No reference source is
 * modified or copied. */
#include <stdint.h>
void __real_ec_enc_init(void *e, unsigned char *buf, uint32_t size);
void __wrap_ec_enc_init(void *e, unsigned char *buf, uint32_t size)
{
    __real_ec_enc_init(e, buf, size);
    *(int *)((char *)e + 52) = 1;   /* ec_ctx.EC_DIFF: after buf(8) storage,end_offs,end_window(12) nend_bits,nbits_total(8) offs,rng,val,ext(16) rem,error(8) */
}

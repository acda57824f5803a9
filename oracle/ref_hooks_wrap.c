/* oracle/ref_hooks_wrap.c -- TEST INFRASTRUCTURE ONLY.
 * Two of the reference's hookable kernels are `static` in their source files (exp_rotation1, celt/vq.c:42-68;
 * comb_filter_const_c, celt/celt.c:156-181, static unless a platform override is compiled in), so the compiled
 * reference library does not export them. This file is compiled twice by oracle/Makefile, each time including ONE
 * reference source file by path (in place, nothing copied) and adding an exported trampoline next to the static
 * function; the two objects go into oracle/_ref/librefhooks.so, which tests/test_hooks_gpu.py calls as the checker. */
#include REF_SRC

#if defined(WRAP_VQ)
void refhook_exp_rotation1(celt_norm *X, int len, int stride, int c, int s)
{
    exp_rotation1(X, len, stride, (opus_val16)c, (opus_val16)s);
}
#elif defined(WRAP_CELT)
void refhook_comb_filter_const(opus_val32 *y, opus_val32 *x, int T, int N, int g10, int g11, int g12)
{
    comb_filter_const_c(y, x, T, N, (opus_val16)g10, (opus_val16)g11, (opus_val16)g12);
}
#endif

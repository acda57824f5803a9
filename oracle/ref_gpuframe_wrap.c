/* oracle/ref_gpuframe_wrap.c -- TEST INFRASTRUCTURE ONLY.
 * Linked with -Wl,--wrap=silk_encode_frame_FIX,--wrap=silk_VAD_GetSA_Q8_c into a variant of the compiled reference
 * (oracle/_ref/libopus_ref_gpuframe.so): the UNMODIFIED reference encoder whose SILK frame function -- and, optionally, its voice
 * activity detector -- is redirected at link level to the hooks of the same argument list in libopusgpu.so
 * (include/opusgpu_hooks.h: opusgpu_silk_encode_frame_FIX, opusgpu_silk_VAD_GetSA_Q8_c). This is the integration a maintainer would
 * make; tests/test_hooks_gpu.py encodes the same PCM through this library and through the plain reference and
 * compares the packets. Until refgpu_load() has been called the wraps forward to the reference's own functions. */
#include <dlfcn.h>
#include <stddef.h>
#include "main_FIX.h"

static void *g_lib;
static int (*g_frame)(void *, opus_int32 *, void *, int, int, int);
static int (*g_vad)(void *, const opus_int16 *);
static int (*g_last_error)(void);
static int g_mask, g_calls[2], g_failures, g_first_error, g_fallbacks;
#define OPUSGPU_UNIMPLEMENTED_ (-5)            /* include/opusgpu.h: the hook left *psEnc untouched */

/* mask: 1 = silk_encode_frame_FIX, 2 = silk_VAD_GetSA_Q8_c. Returns 0 on success. */
int refgpu_load(const char *libopusgpu_path, int mask)
{
    g_lib = dlopen(libopusgpu_path, RTLD_NOW | RTLD_LOCAL);
    if (!g_lib) return -1;
    g_frame = (int (*)(void *, opus_int32 *, void *, int, int, int))dlsym(g_lib, "opusgpu_silk_encode_frame_FIX");
    g_vad = (int (*)(void *, const opus_int16 *))dlsym(g_lib, "opusgpu_silk_VAD_GetSA_Q8_c");
    g_last_error = (int (*)(void))dlsym(g_lib, "opusgpu_get_last_error");
    if (!g_frame || !g_vad || !g_last_error) return -2;
    g_mask = mask; g_calls[0] = g_calls[1] = g_failures = g_first_error = g_fallbacks = 0;
    return 0;
}
int refgpu_calls(int which) { return g_calls[which]; }
int refgpu_failures(void) { return g_failures; }
int refgpu_fallbacks(void) { return g_fallbacks; }      /* frames the hook declined (UNIMPLEMENTED) and the reference coded */
int refgpu_first_error(void) { return g_first_error; }

opus_int __real_silk_encode_frame_FIX(silk_encoder_state_FIX *psEnc, opus_int32 *pnBytesOut, ec_enc *psRangeEnc, opus_int condCoding, opus_int maxBits, opus_int useCBR);
opus_int __wrap_silk_encode_frame_FIX(silk_encoder_state_FIX *psEnc, opus_int32 *pnBytesOut, ec_enc *psRangeEnc, opus_int condCoding, opus_int maxBits, opus_int useCBR)
{
    if (!(g_mask & 1)) return __real_silk_encode_frame_FIX(psEnc, pnBytesOut, psRangeEnc, condCoding, maxBits, useCBR);
    g_calls[0]++;
    const int ret = g_frame(psEnc, pnBytesOut, psRangeEnc, condCoding, maxBits, useCBR);
    if (ret != 0 && g_last_error() == OPUSGPU_UNIMPLEMENTED_) {
        /* outside the hook's operating region (12 kHz, a bandwidth-transition frame, LBRR): it promises not to have touched
         * the state in that case, so the frame goes through the reference's own function. The caller (silk/enc_API.c:499)
         * only silk_assert()s the return value: without this fall-back a release build would emit garbage for such a frame. */
        g_fallbacks++;
        return __real_silk_encode_frame_FIX(psEnc, pnBytesOut, psRangeEnc, condCoding, maxBits, useCBR);
    }
    if (ret != 0) { if (!g_failures) g_first_error = g_last_error(); g_failures++; }
    return ret;
}

opus_int __real_silk_VAD_GetSA_Q8_c(silk_encoder_state *psEncC, const opus_int16 pIn[]);
opus_int __wrap_silk_VAD_GetSA_Q8_c(silk_encoder_state *psEncC, const opus_int16 pIn[])
{
    if (!(g_mask & 2)) return __real_silk_VAD_GetSA_Q8_c(psEncC, pIn);
    g_calls[1]++;
    const int ret = g_vad(psEncC, pIn);
    if (ret != 0) { if (!g_failures) g_first_error = g_last_error(); g_failures++; }
    return ret;
}

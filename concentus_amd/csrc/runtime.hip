// runtime.hip -- process-level plumbing of libopusgpu.so: version, error strings, device queries.
#include <stdio.h>
#include "opusgpu_internal.h"

static thread_local int g_last_error = OPUSGPU_OK;

extern "C" void opusgpu_set_last_error(int err) { g_last_error = err; }
extern "C" int opusgpu_get_last_error(void) { return g_last_error; }

extern "C" int opusgpu_check_launch(void)
{
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) return OPUSGPU_OK;
    fprintf(stderr, "opusgpu: HIP error: %s\n", hipGetErrorString(e));
    return OPUSGPU_INTERNAL_ERROR;
}

extern "C" const char *opusgpu_get_version_string(void) { return "opusgpu 0.1.0 gfx950 (libopus 1.1.2-fixed bitstream)"; }

extern "C" const char *opusgpu_strerror(int error)
{
    static const char *const msg[8] = {
        "success", "invalid argument", "buffer too small", "internal error",
        "corrupted stream", "request not implemented", "invalid state", "memory allocation failed"};
    if (error > 0 || error < -7) return "unknown error";
    return msg[-error];
}

extern "C" int opusgpu_num_cus(void)
{
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 256;
        cus = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
    }
    return cus;
}

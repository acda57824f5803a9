// runtime.hip -- process-level plumbing of libopusgpu.so: version, error strings, device queries.
#include <stdio.h>
#include <mutex>
#include <vector>
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

// ---- optional per-kernel timing (HIP events on the launch stream) ----
namespace {
struct TimedLaunch { int kernel; hipEvent_t t0, t1; };
std::mutex g_tm;
bool g_timing = false;
std::vector<TimedLaunch> g_launches;
}

extern "C" int opusgpu_kernel_timing_enable(int on)
{
    std::lock_guard<std::mutex> lk(g_tm);
    g_timing = on != 0;
    return OPUSGPU_OK;
}

// internal: bracket one launch. begin returns a slot index or -1 when timing is off.
extern "C" int opusgpu_timing_begin(int kernel, hipStream_t s)
{
    std::lock_guard<std::mutex> lk(g_tm);
    if (!g_timing) return -1;
    TimedLaunch t;
    t.kernel = kernel;
    if (hipEventCreate(&t.t0) != hipSuccess || hipEventCreate(&t.t1) != hipSuccess) return -1;
    hipEventRecord(t.t0, s);
    g_launches.push_back(t);
    return (int)g_launches.size() - 1;
}

extern "C" void opusgpu_timing_end(int slot, hipStream_t s)
{
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_tm);
    hipEventRecord(g_launches[slot].t1, s);
}

extern "C" int opusgpu_kernel_timing_read(double *ms_sum, int *launches, int n_kernels)
{
    std::lock_guard<std::mutex> lk(g_tm);
    if (!ms_sum || !launches || n_kernels < 0) return OPUSGPU_BAD_ARG;
    for (int k = 0; k < n_kernels; k++) { ms_sum[k] = 0; launches[k] = 0; }
    int rc = OPUSGPU_OK;
    for (auto &t : g_launches) {
        float ms = 0;
        if (hipEventSynchronize(t.t1) != hipSuccess || hipEventElapsedTime(&ms, t.t0, t.t1) != hipSuccess)
            rc = OPUSGPU_INTERNAL_ERROR;
        else if (t.kernel >= 0 && t.kernel < n_kernels) { ms_sum[t.kernel] += ms; launches[t.kernel]++; }
        hipEventDestroy(t.t0);
        hipEventDestroy(t.t1);
    }
    g_launches.clear();
    return rc;
}

// ---- bad-record counter of the SILK record kernels (silk_validate.h) ----
// One counter per device, allocated on first use and never freed (4 bytes for the life of the process).
// The one-record hooks count into a counter of their own thread instead (opusgpu_private_bad_counter_begin/_end below),
// so a hook call never reads or clears what a batch running on another thread has counted.
namespace {
std::mutex g_bad_m;
int *g_bad[64] = {};
thread_local int *t_bad_private[64] = {};
thread_local int *t_bad_active = nullptr;
}

extern "C" int *opusgpu_bad_record_counter(void)
{
    if (t_bad_active) return t_bad_active;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lk(g_bad_m);
    if (!g_bad[dev]) {
        int *p = nullptr;
        if (hipMalloc((void **)&p, sizeof(int)) != hipSuccess) return nullptr;
        if (hipMemset(p, 0, sizeof(int)) != hipSuccess) { (void)hipFree(p); return nullptr; }
        g_bad[dev] = p;
    }
    return g_bad[dev];
}

// From here to opusgpu_private_bad_counter_end() the record kernels launched by THIS thread count into a counter only this
// thread reads (4 bytes per thread and device, allocated on first use, never freed). Returns OPUSGPU_OK or an error code.
extern "C" int opusgpu_private_bad_counter_begin(void)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return OPUSGPU_INTERNAL_ERROR;
    if (!t_bad_private[dev] && hipMalloc((void **)&t_bad_private[dev], sizeof(int)) != hipSuccess) { t_bad_private[dev] = nullptr; return OPUSGPU_ALLOC_FAIL; }
    if (hipMemset(t_bad_private[dev], 0, sizeof(int)) != hipSuccess) return OPUSGPU_INTERNAL_ERROR;
    t_bad_active = t_bad_private[dev];
    return OPUSGPU_OK;
}

// Ends the scope; returns the records rejected inside it (after the null stream has drained), or a negative error code.
extern "C" int opusgpu_private_bad_counter_end(void)
{
    int *p = t_bad_active;
    t_bad_active = nullptr;
    if (!p) return 0;
    int n = 0;
    if (hipMemcpy(&n, p, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return OPUSGPU_INTERNAL_ERROR;
    return n;
}

// Records rejected by the SILK batch kernels of the current device since the last call (waits for `stream`).
extern "C" int opusgpu_silk_bad_records(void *stream)
{
    int *p = opusgpu_bad_record_counter();
    if (!p) return OPUSGPU_ALLOC_FAIL;
    int n = 0;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemcpyAsync(&n, p, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipMemsetAsync(p, 0, sizeof(int), s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
        return OPUSGPU_INTERNAL_ERROR;
    return n;
}

// Records per wavefront of the scratch-resident SILK analysis kernels (silk_pitch_kernels.hip, silk_nlsf_kernels.hip). Read per
// call so that tests and sweeps can compare the mappings.
extern "C" int opusgpu_silk_lanes_per_block(void)
{
    const char *e = getenv("OPUSGPU_SILK_LANES");
    const int v = e ? atoi(e) : 64;
    return (v == 8 || v == 16 || v == 32 || v == 64) ? v : 64;
}

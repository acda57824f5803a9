// wave.h -- the execution abstraction the frame kernels are written against.
//
// On gfx950 one 64-lane wavefront cooperates on one Opus frame: data-parallel loops are strided over
// the lanes (`for (i = lane(); i < n; i += LANES)`), scalar control code runs redundantly (and
// identically) on all lanes, and `wave_sync()` orders LDS traffic between the lanes of the wave.
//
// With -DCA_HOST_EMU the same sources compile as plain C++ with LANES == 1 (every collective becomes
// the identity). That build exists ONLY so tests can single-step the kernel logic on a CPU
// (tests/emu); it is not a product path and nothing in concentus_amd/ loads it.
#pragma once
#include <stdint.h>

// CA_LANE_FRAME: device build in which ONE LANE owns one frame (LANES == 1 semantics on the GPU): the
// same sources, every collective the identity, the per-frame working set in private memory.
#if defined(CA_HOST_EMU) || defined(CA_LANE_FRAME)
#define CA_SINGLE_LANE 1
#endif

#if defined(CA_HOST_EMU)
#include <stdio.h>
#if defined(CA_HOST_EMU_TRACE)
#define CA_TRACE(...) do { fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); } while (0)
#else
#define CA_TRACE(...) do {} while (0)
#endif
// debug taps (host emulation only): record a named snapshot of an intermediate array
extern "C" void emu_tap(const char *name, const void *p, int bytes);
#define CA_TAP(name, p, bytes) emu_tap(name, p, bytes)
// workload counters (host emulation only): sizes the per-frame work of a stage for the roofline notes
extern "C" void emu_count(const char *name, long n);
#define CA_COUNT(name, n) emu_count(name, (long)(n))
#define CA_DEV static inline
#define CA_HOSTDEV static inline
#define CA_DEVICE_CONST static const
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
static inline int2 make_int2(int x, int y) { int2 v = {x, y}; return v; }
static inline int __mul24(int a, int b)
{
    int32_t sa = (int32_t)((uint32_t)a << 8) >> 8, sb = (int32_t)((uint32_t)b << 8) >> 8;
    return (int32_t)((uint32_t)sa * (uint32_t)sb);
}
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
#else
#include <hip/hip_runtime.h>
#define CA_DEV __device__ __forceinline__
#define CA_HOSTDEV __host__ __device__ __forceinline__
#define CA_DEVICE_CONST static __device__ const
#define CA_TRACE(...) do {} while (0)
#define CA_TAP(name, p, bytes) do {} while (0)
#define CA_COUNT(name, n) do {} while (0)
#endif

// Stage stamps: only in the diagnostic kernel build (-DCA_STAGE_TIMING, celt_enc_kernel_diag.hip); the
// stamp values go to a buffer of their own and feed no output (cdna_hip_programming.md, "In-kernel stamps").
#if defined(CA_STAGE_TIMING) && !defined(CA_HOST_EMU)
namespace ca { struct StageClock { unsigned long long *acc; unsigned long long last; }; }
#define CA_STAMP_DECL ca::StageClock *stage_clock
#define CA_STAMP_F(F, k) do { ca::StageClock *stage_clock = (ca::StageClock *)(F).diag; CA_STAMP(k); } while (0)
// (sched_barrier: the instruction scheduler may otherwise move a stage's arithmetic across the clock reads, and the shares lie)
#define CA_STAMP(k) do { if (stage_clock && stage_clock->acc) { __builtin_amdgcn_sched_barrier(0); unsigned long long _t = __builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_s_waitcnt(0xC07F); stage_clock->acc[k] += _t - stage_clock->last; stage_clock->last = __builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
namespace ca { struct StageClock; }
#define CA_STAMP(k) do {} while (0)
#define CA_STAMP_F(F, k) do {} while (0)
#endif

// Lane-strided loops become plain serial loops in the lane-per-frame build; unrolling them lets the loads of
// several iterations be in flight together (the lone wavefront of a SIMD has nothing else to hide them with).
#if defined(CA_LANE_FRAME)
#define CA_UNROLL_LANE _Pragma("unroll 8")
#else
#define CA_UNROLL_LANE
#endif

namespace ca {

#if defined(CA_SINGLE_LANE)
enum { LANES = 1 };
CA_DEV int lane() { return 0; }
// still a compiler barrier: the working-set structs are read through differently typed views (unions,
// i16 views of i32 arrays) and the sources rely on wave_sync() to order those accesses
CA_DEV void wave_sync() { asm volatile("" ::: "memory"); }
template <class T> CA_DEV T shfl_xor(T v, int) { return v; }
template <class T> CA_DEV T bcast(T v, int) { return v; }
#else
enum { LANES = 64 };
CA_DEV int lane() { return (int)(threadIdx.x & 63); }
// LDS operations of one wave execute in issue order, so ordering between its lanes only needs the
// compiler fenced (verified in the ISA: no s_barrier, no extra waitcnt).
CA_DEV void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <class T> CA_DEV T shfl_xor(T v, int m) { return __shfl_xor(v, m, 64); }
template <class T> CA_DEV T bcast(T v, int src) { return __shfl(v, src, 64); }
#endif

// ---- all-lanes reductions (result valid, and identical, in every lane) -----------------------------
#if defined(CA_SINGLE_LANE)
CA_DEV int32_t wave_add(int32_t v) { return v; }
CA_DEV int32_t wave_max(int32_t v) { return v; }
CA_DEV int32_t wave_min(int32_t v) { return v; }
CA_DEV uint32_t wave_or(uint32_t v) { return v; }
CA_DEV int64_t wave_add64(int64_t v) { return v; }
CA_DEV int32_t uni(int32_t v) { return v; }
CA_DEV uint32_t uni(uint32_t v) { return v; }
CA_DEV int32_t wave_scan_add(int32_t v) { return v; }
CA_DEV int32_t wave_last(int32_t v) { return v; }
CA_DEV uint64_t wave_ballot(bool p) { return p ? 1u : 0u; }
CA_DEV int32_t lane_bcast(int32_t v, int) { return v; }
#else
// DPP row operations instead of ds_bpermute shuffles: an inclusive scan over the 64 lanes in six VALU
// steps (row_shr 1/2/4/8 inside each row of 16, row_bcast15 into rows 1/3, row_bcast31 into the upper
// half), after which lane 63 holds the reduction; v_readlane_b32 returns it as a wave-uniform SGPR.
// (gfx9 DPP controls: row_shr:n = 0x110+n, row_bcast15 = 0x142, row_bcast31 = 0x143.)
#define CA_DPP(old, v, ctrl, rowmask) __builtin_amdgcn_update_dpp((old), (v), (ctrl), (rowmask), 0xf, false)
#define CA_WAVE_REDUCE(v, IDENT, OP)                         \
    do {                                                      \
        int32_t _t;                                           \
        _t = CA_DPP(IDENT, v, 0x111, 0xf); v = OP(v, _t);     \
        _t = CA_DPP(IDENT, v, 0x112, 0xf); v = OP(v, _t);     \
        _t = CA_DPP(IDENT, v, 0x114, 0xf); v = OP(v, _t);     \
        _t = CA_DPP(IDENT, v, 0x118, 0xf); v = OP(v, _t);     \
        _t = CA_DPP(IDENT, v, 0x142, 0xa); v = OP(v, _t);     \
        _t = CA_DPP(IDENT, v, 0x143, 0xc); v = OP(v, _t);     \
        v = __builtin_amdgcn_readlane(v, 63);                 \
    } while (0)
#define CA_OP_ADD(a, b) ((int32_t)((uint32_t)(a) + (uint32_t)(b)))
#define CA_OP_MAX(a, b) ((a) > (b) ? (a) : (b))
#define CA_OP_MIN(a, b) ((a) < (b) ? (a) : (b))
#define CA_OP_OR(a, b) ((a) | (b))
CA_DEV int32_t wave_add(int32_t v) { CA_WAVE_REDUCE(v, 0, CA_OP_ADD); return v; }
CA_DEV int32_t wave_max(int32_t v) { CA_WAVE_REDUCE(v, (int32_t)0x80000000, CA_OP_MAX); return v; }
CA_DEV int32_t wave_min(int32_t v) { CA_WAVE_REDUCE(v, 0x7fffffff, CA_OP_MIN); return v; }
CA_DEV uint32_t wave_or(uint32_t u) { int32_t v = (int32_t)u; CA_WAVE_REDUCE(v, 0, CA_OP_OR); return (uint32_t)v; }
CA_DEV int64_t wave_add64(int64_t v)
{
    for (int m = LANES >> 1; m > 0; m >>= 1) {
        uint32_t lo = shfl_xor((uint32_t)v, m), hi = shfl_xor((uint32_t)((uint64_t)v >> 32), m);
        v = (int64_t)((uint64_t)v + (((uint64_t)hi << 32) | lo));
    }
    return v;
}
// Assert to the compiler that a value is wave-uniform (it then lives in an SGPR and feeds scalar
// branches / scalar loads).
// inclusive prefix sum over the lanes (the scan the reductions are built on), value of the last lane,
// ballot, and broadcast of one lane's value (src uniform)
CA_DEV int32_t wave_scan_add(int32_t v)
{
    int32_t _t;
    _t = CA_DPP(0, v, 0x111, 0xf); v = CA_OP_ADD(v, _t);
    _t = CA_DPP(0, v, 0x112, 0xf); v = CA_OP_ADD(v, _t);
    _t = CA_DPP(0, v, 0x114, 0xf); v = CA_OP_ADD(v, _t);
    _t = CA_DPP(0, v, 0x118, 0xf); v = CA_OP_ADD(v, _t);
    _t = CA_DPP(0, v, 0x142, 0xa); v = CA_OP_ADD(v, _t);
    _t = CA_DPP(0, v, 0x143, 0xc); v = CA_OP_ADD(v, _t);
    return v;
}
CA_DEV int32_t wave_last(int32_t v) { return __builtin_amdgcn_readlane(v, 63); }
CA_DEV uint64_t wave_ballot(bool p) { return __ballot(p); }
CA_DEV int32_t lane_bcast(int32_t v, int src) { return __builtin_amdgcn_readlane(v, src); }
CA_DEV int32_t uni(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }
CA_DEV uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)v); }
#endif

}  // namespace ca

// Address spaces of the lane-per-frame build. A per-frame array there lives in one of three places -- the workgroup's
// LDS scratch ([element][lane], one column per lane), this lane's private memory, or the frame's rows in HBM -- and
// the pointer TYPE says which: a generic pointer (the struct member reloaded from private memory has lost its
// provenance) makes the compiler emit FLAT loads/stores, which take the slow aperture path for LDS, count as vector
// memory AND as LDS operations and force s_waitcnt vmcnt(0) lgkmcnt(0) on every use. With the address space in the
// type the same accesses are ds_read/ds_write, scratch_* and global_* instructions.
//   x16_t      element of the frame's normalised-band array X: HBM (address space 1) in the lane build
//   LdsCol<T>  one lane's column of an [element][64] LDS array (address space 3, stride 64 elements)
//   Priv<T>    an array in this lane's private memory (stride 1)
#if defined(CA_LANE_FRAME)
// (with CA_HOST_EMU as well -- tests/emu/celt_lane_emu.cpp, host clang -- the qualifiers stay in the types and mean nothing to
// the x86 back end: the lane build's own code paths run on a CPU, one "lane" of a 64-column LDS image)
#if defined(CA_HOST_EMU)
#define CA_MEMBER inline
#else
#define CA_MEMBER __device__ __forceinline__
#endif
#define CA_AS_LDS __attribute__((address_space(3)))
#define CA_AS_GLB __attribute__((address_space(1)))
namespace ca {
typedef CA_AS_GLB int16_t x16_t;
typedef int v4i __attribute__((ext_vector_type(4)));      // 16 bytes moved by one instruction through an address-space-qualified pointer
template <class T> struct LdsCol {
    CA_AS_LDS T *p;
    CA_MEMBER CA_AS_LDS T &operator[](int j) const { return p[j * 64]; }
    CA_MEMBER LdsCol operator+(int o) const { LdsCol r; r.p = p + o * 64; return r; }
};
template <class T> CA_MEMBER LdsCol<T> lds_col(CA_AS_LDS T *p) { LdsCol<T> r; r.p = p; return r; }
template <class T> struct Priv {
    T *p;
    CA_MEMBER T &operator[](int j) const { return p[j]; }
    CA_MEMBER Priv operator+(int o) const { Priv r; r.p = p + o; return r; }
};
template <class T> CA_MEMBER Priv<T> priv(T *p) { Priv<T> r; r.p = p; return r; }
}
#else
namespace ca { typedef int16_t x16_t; }
#endif

namespace ca {
// LP<T>: pointer to a per-frame array (decoder sources). In the lane-per-frame build the hot little arrays live in LDS laid out
// [element][lane] (stride 64), the rest stays contiguous (stride 1): one runtime stride covers both. In the
// other builds it is a plain pointer.
#if defined(CA_LANE_FRAME)
template <class T> struct LP {
    T *p;
    int s;
    CA_MEMBER T &operator[](int j) const { return p[j * s]; }
    CA_MEMBER LP operator+(int o) const { LP r; r.p = p + o * s; r.s = s; return r; }
};
template <class T> CA_MEMBER LP<T> lp_make(T *p, int stride) { LP<T> r; r.p = p; r.s = stride; return r; }
#else
template <class T> using LP = T *;
template <class T> CA_DEV LP<T> lp_make(T *p, int) { return p; }
#endif

}  // namespace ca

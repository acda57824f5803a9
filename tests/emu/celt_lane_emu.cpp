// tests/emu/celt_lane_emu.cpp -- TEST INFRASTRUCTURE: host build of the LANE-PER-FRAME variant of the frame-kernel sources
// (CA_LANE_FRAME + CA_HOST_EMU, compiled with the image's clang for x86). The back phase and the decoder's first stage have code
// paths of their own in that variant (per-lane LDS columns laid out [slot][64], typed address spaces, the leaf quantiser written
// for one lane, the per-lane working set of celt_back_lane_kernel); this library runs exactly those paths on a CPU, one frame at
// a time in a chosen column of a 64-column LDS image, so that the goldens and the fuzz sweeps cover them without a GPU.
// Not a product path: concentus_amd/ never loads this library.
#define CA_HOST_EMU 1
#define CA_LANE_FRAME 1
#include <stdlib.h>
#include <string.h>
#include <map>
#include <string>
extern "C" void emu_tap(const char *, const void *, int) {}
// event counters (CA_COUNT): the tests check that the lane build's rare paths are actually reached
static std::map<std::string, std::pair<long, long>> g_counts;
extern "C" void emu_count(const char *name, long n) { auto &c = g_counts[name]; c.first++; c.second += n; }
extern "C" void emu_lane_counts_reset(void) { g_counts.clear(); }
extern "C" long emu_lane_count(const char *name) { auto it = g_counts.find(name); return it == g_counts.end() ? 0 : it->second.first; }
#define CA_LANE_SLOTS 264                  // as celt_back_lane_kernel.hip (the decoder part uses the first 240)
#include "../../concentus_amd/csrc/celt_lane_tables.h"
#include "../../concentus_amd/csrc/celt_enc.h"
#include "../../concentus_amd/csrc/celt_dec.h"

using namespace ca;

// slot: the column (0..63) of the workgroup's LDS image this "lane" owns; neighbours are poisoned and checked afterwards
static int g_slot = 0;
extern "C" void emu_lane_set_slot(int s) { g_slot = s & 63; }

static void lds_poison() { memset(g_lds_scratch, 0x5A, sizeof(g_lds_scratch)); }
// n16 = number of leading 16-bit slots of the image ([slot][lane], 2-byte elements); what follows is laid out in 32-bit slots
// (the decoder's pulse vector): column c of a 32-bit row occupies the 16-bit cells 2c and 2c + 1 of that 128-cell row
static int lds_neighbours_untouched(int n16)
{
    const int cells = (int)(sizeof(g_lds_scratch) / sizeof(g_lds_scratch[0]));
    for (int k = 0; k < cells; k++) {
        const int col = k < n16 * 64 ? k % 64 : ((k - n16 * 64) % 128) / 2;
        if (col != g_slot && g_lds_scratch[k] != 0x5A5A) return 0;
    }
    return 1;
}

extern "C" int emu_lane_celt_encode_frames(const opusgpu_celt_config *cfg, opusgpu_celt_state *states /* or NULL */,
                                           const int16_t *pcm, int nframes, int frames_per_stream,
                                           unsigned char *out, int out_stride, int *out_len, uint32_t *out_rng)
{
    // front phase: the wave-per-frame sources with one lane (as in libcelt_emu.so); back phase: celt_back_lane_kernel's body
    FrontLds *F1 = (FrontLds *)aligned_alloc(64, sizeof(FrontLds) + 64);
    FrameMid *mid = (FrameMid *)aligned_alloc(64, sizeof(FrameMid) + 64);
    const int C = cfg->channels;
    int clean = 1;
    fill_lds_tables();
    for (int n = 0; n < nframes; n++) {
        memset(F1, 0xAB, sizeof(FrontLds));
        memset(mid, 0xCD, sizeof(FrameMid));
        opusgpu_celt_state *st = states ? &states[n / frames_per_stream] : NULL;
        celt_encode_front(*F1, *cfg, st, st, pcm + (size_t)n * 960 * C, mid);
        lds_poison();
        BackLds F;
        memset(&F, 0xAB, sizeof(F));
        F.col = (CA_AS_LDS i16 *)(g_lds_scratch + g_slot);
        FrameResult r = celt_encode_back(F, *cfg, mid, st, out + (size_t)n * out_stride);
        out_len[n] = r.bytes;
        out_rng[n] = r.final_range;
        clean &= lds_neighbours_untouched(LS_SLOTS);
    }
    free(F1);
    free(mid);
    return clean ? 0 : -1;                    // -1: the lane wrote outside its own LDS column
}

static void dec_state_reset(opusgpu_celt_dec_state *st)
{
    memset(st, 0, sizeof(*st));
    for (int i = 0; i < 42; i++) st->oldLogE[i] = st->oldLogE2[i] = -28672;
}

extern "C" int emu_lane_celt_decode_frames(const unsigned char *packets, int stride, const int *len, int nframes,
                                           int frames_per_stream, int16_t *pcm, uint32_t *rng, int *ret)
{
    SynthLds *L = (SynthLds *)aligned_alloc(64, sizeof(SynthLds) + 64);
    opusgpu_celt_dec_state *st = (opusgpu_celt_dec_state *)aligned_alloc(64, sizeof(opusgpu_celt_dec_state) + 64);
    int clean = 1;
    fill_lds_tables();
    for (int n = 0; n < nframes; n++) {
        if (n % frames_per_stream == 0) dec_state_reset(st);
        lds_poison();
        DecWork F;
        memset(&F, 0xAB, sizeof(F));
        F.lds_iy16 = (CA_AS_LDS i16 *)(g_lds_iy16 + g_slot);
        F.lds_pvq16 = (CA_AS_LDS i16 *)(g_lds_pvq16 + g_slot);
        memset(L, 0xAB, sizeof(SynthLds));
        DecResult r = celt_decode_frame(F, *L, st, packets + (size_t)n * stride, len[n], pcm + (size_t)n * 960 * 2);
        ret[n] = r.samples;
        rng[n] = r.final_range;
        clean &= lds_neighbours_untouched(LS_SLOTS);          // (the decoder's part of the image is 16-bit slots throughout since round 3)
    }
    free(L);
    free(st);
    return clean ? 0 : -1;
}

// the lane build's pulse-cache look-ups (celt_enc_back.h: two rounds of probes, row length / last entry from derived tables) and
// the reference's bisection (rate.h:51-77) on the tables as they are, for tests/test_lane_emu_cpu.py
extern "C" int emu_lane_bits2pulses(int band, int LM, int bits) { fill_lds_tables(); return bits2pulses(band, LM, bits); }
extern "C" int emu_lane_pulse_cache_max(int band, int LM) { fill_lds_tables(); return pulse_cache_max(band, LM); }
extern "C" int emu_lane_bits2pulses_bisect(int band, int LM, int bits)
{
    fill_lds_tables();
    const u8 *cache = pulse_cache(band, LM);
    int lo = 0, hi = cache[0];
    bits--;
    for (int i = 0; i < 6; i++) {
        int mid = (lo + hi + 1) >> 1;
        if ((int)cache[mid] >= bits) hi = mid; else lo = mid;
    }
    if (bits - (lo == 0 ? -1 : (int)cache[lo]) <= (int)cache[hi] - bits) return lo;
    return hi;
}
extern "C" int emu_lane_pulse_cache_max_ref(int band, int LM) { fill_lds_tables(); const u8 *cache = pulse_cache(band, LM); return cache[cache[0]]; }

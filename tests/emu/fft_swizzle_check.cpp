// tests/emu/fft_swizzle_check.cpp -- TEST INFRASTRUCTURE: the bank swizzle of the FFT scratch (mdct_dev.h fsw<0>) is a bijection of [0, 480), the
// per-butterfly member addressing (fft_members) and the pre-swizzled bit-reversal table agree with it everywhere. Built and run by tests/test_fft_swizzle_cpu.py.
#define CA_HOST_EMU 1
#include <stdio.h>
extern "C" void emu_tap(const char*, const void*, int) {}
extern "C" void emu_count(const char*, long) {}
#include "../../concentus_amd/csrc/mdct_dev.h"
using namespace ca;
int main() {
    int bad = 0;
    for (int e = 0; e < 480; e += 1) {
        if ((e & 31) < 8) { int a[4]; fft_members<0, 4, 8>(e, a); for (int c = 0; c < 4; c++) if (a[c] != fsw<0>(e + 8 * c)) bad++; }
        if (e % 96 < 32) { int a[3]; fft_members<0, 3, 32>(e, a); for (int c = 0; c < 3; c++) if (a[c] != fsw<0>(e + 32 * c)) bad++; }
        if (e < 96) { int a[5]; fft_members<0, 5, 96>(e, a); for (int c = 0; c < 5; c++) if (a[c] != fsw<0>(e + 96 * c)) bad++; }
    }
    for (int g = 0; g < 60; g++) for (int q = 0; q < 4; q++) if ((fsw<0>(8 * g) ^ (2 * q)) != fsw<0>(8 * g + 2 * q)) bad++;
    for (int i = 0; i < 480; i++) if (CLT_fft_bitrev480_sw[i] != fsw<0>(CLT_fft_bitrev480[i])) bad++;
    // bijection
    int seen[480] = {0}; for (int e = 0; e < 480; e++) { int p = fsw<0>(e); if (p < 0 || p >= 480 || seen[p]++) bad++; }
    printf("mismatches %d\n", bad);
    return bad != 0;
}

// celt_enc_back.h -- bit allocation, fine energy, and the PVQ band quantiser (encoder side).
//
// Wave-cooperative counterparts of:
//   compute_allocation / interp_bits2pulses      opus-fix/celt/rate.c:527-639, :248-525; init_caps celt/celt.c:246-256
//   quant_fine_energy / quant_energy_finalise    celt/quant_bands.c:369-404, :406-439
//   quant_all_bands (encode = 1)                 celt/bands.c:1337-1502
//     quant_band / quant_band_stereo / quant_partition / quant_band_n1 / compute_theta / compute_qn
//                                                celt/bands.c:1044-1174, :1176-1335, :864-1042, :819-862, :645-817, :596-621
//     deinterleave_hadamard / haar1 / intensity_stereo / stereo_split   celt/bands.c:524-549, :581-594, :336-360, :362-373
//   alg_quant / exp_rotation / exp_rotation1 / stereo_itheta            celt/vq.c:161-325, :70-117, :43-68, :376-408
//   encode_pulses / icwrs                        celt/cwrs.c:440-460
//
// Encoder-only simplifications that follow from `resynth == 0` in the reference (bands.c:1046-1050):
// lowband folding, `norm`, `fill`, collapse masks, `gain` and the LCG `seed` never influence the coded
// bits, so they are not carried here (the decoder path will need them).
#pragma once
#include "celt_enc_mid.h"
#include "rangedec.h"

namespace ca {

enum { PVQ_LDS_N = 48 };       // lane build: largest leaf whose search state lives in LDS
enum { BITRES = 3, ALLOC_STEPS = 6, MAX_FINE_BITS = 8, FINE_OFFSET = 21, QTHETA_OFFSET = 4, QTHETA_OFFSET_TWOPHASE = 16 };

CA_DEV int get_pulses(int i) { return i < 8 ? i : (8 + (i & 7)) << ((i >> 3) - 1); }     // rate.h:46-49

CA_DEV const u8 *pulse_cache(int band, int LM) { return CLT_cache_bits50 + CLT_cache_index50[(LM + 1) * NB + band]; }

#if defined(CA_LANE_FRAME)
// Lane build: every look-up is an LDS round trip on the critical path of a lone wavefront, so the chain is kept short -- the
// row's last entry and length come from tables of their own (celt_lane_tables.h), and the row (non-decreasing, at most 40
// entries long) is searched in two rounds of independent probes instead of six dependent bisection steps.
CA_DEV int pulse_cache_max(int band, int LM) { return g_lds_tables.cache_max_[(LM + 1) * NB + band]; }

CA_DEV int bits2pulses(int band, int LM, int bits)                                         // rate.h:51-77
{
    const u8 *cache = pulse_cache(band, LM);
    const int mx = g_lds_tables.cache_len_[(LM + 1) * NB + band];
    bits--;
    // lo = the largest index whose entry is below `bits` (0: none) = what the reference's bisection ends on; probes 6, 12, .., 36,
    // then the five entries after the last probe that was below
    int v[6], c1 = 0;
#pragma unroll
    for (int t = 0; t < 6; t++) v[t] = cache[imin(6 * (t + 1), mx)];
#pragma unroll
    for (int t = 0; t < 6; t++) c1 += (6 * (t + 1) <= mx) & (v[t] < bits);
    int w[5], c2 = 0;
#pragma unroll
    for (int u = 0; u < 5; u++) w[u] = cache[imin(6 * c1 + u + 1, mx)];
#pragma unroll
    for (int u = 0; u < 5; u++) c2 += (6 * c1 + u + 1 <= mx) & (w[u] < bits);
    const int lo = 6 * c1 + c2, hi = lo < mx ? lo + 1 : mx;
    if (bits - (lo == 0 ? -1 : (int)cache[lo]) <= (int)cache[hi] - bits) return lo;
    return hi;
}
#else
CA_DEV int pulse_cache_max(int band, int LM) { const u8 *cache = pulse_cache(band, LM); return cache[cache[0]]; }

CA_DEV int bits2pulses(int band, int LM, int bits)                                         // rate.h:51-77
{
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
#endif

CA_DEV int pulses2bits(int band, int LM, int pulses)                                       // rate.h:79-85
{
    return pulses == 0 ? 0 : pulse_cache(band, LM)[pulses] + 1;
}

CA_DEV u32 pvq_u(int n, int k)                                                             // cwrs.c:197
{
    int lo = n < k ? n : k, hi = n < k ? k : n;
    return CLT_pvq_u_data[CLT_pvq_u_row[lo] + hi];
}

// ---- compute_allocation (rate.c:527-639 + :248-525), start 0, end 21, LM 3, encode 1 ---------------
struct AllocOut { int codedBands; i32 balance; int intensity; int dual_stereo; };

template <class L, class EC>
CA_DEVFN AllocOut compute_allocation_wave(L &F, EC &ec, int C, int alloc_trim, int intensity_in,
                                          int dual_stereo_in, i32 total, int prev, int signalBandwidth)
{
    const int LM = LM3, end = NB, start = 0, len = NB;
    const i16 *eB = CLT_eband5ms;
    i32 *bits = F.pulses, *ebits = F.fine_quant, *fine_priority = F.fine_priority;
    AllocOut out;
    out.intensity = intensity_in;
    out.dual_stereo = dual_stereo_in;
    total = imax(total, 0);
    int skip_start = start;
    int skip_rsv = total >= 1 << BITRES ? 1 << BITRES : 0;
    total -= skip_rsv;
    int intensity_rsv = 0, dual_stereo_rsv = 0;
    if (C == 2) {
        intensity_rsv = CLT_log2_frac_table[end - start];
        if (intensity_rsv > total) intensity_rsv = 0;
        else {
            total -= intensity_rsv;
            dual_stereo_rsv = total >= 1 << BITRES ? 1 << BITRES : 0;
            total -= dual_stereo_rsv;
        }
    }
    CA_UNROLL_LANE
    for (int j = lane(); j < end; j += LANES) {
        int w = eB[j + 1] - eB[j];
        F.thresh[j] = imax(C << BITRES, ((3 * w) << LM << BITRES) >> 4);
        i32 to = (C * w * (alloc_trim - 5 - LM) * (end - j - 1) * (1 << (LM + BITRES))) >> 6;
        if ((w << LM) == 1) to -= C << BITRES;
        F.trim_offset[j] = to;
    }
    wave_sync();
    int lo = 1, hi = 11 - 1;
    do {
        int done = 0, psum = 0, mid = (lo + hi) >> 1;
        for (int j = end; j-- > start;) {
            int N = eB[j + 1] - eB[j];
            int bitsj = (C * N * CLT_band_allocation[mid * len + j]) << LM >> 2;
            if (bitsj > 0) bitsj = imax(0, bitsj + uni(F.trim_offset[j]));
            bitsj += uni(F.offsets[j]);
            if (bitsj >= uni(F.thresh[j]) || done) { done = 1; psum += imin(bitsj, uni(F.cap[j])); }
            else if (bitsj >= C << BITRES) psum += C << BITRES;
        }
        if (psum > total) hi = mid - 1; else lo = mid + 1;
    } while (lo <= hi);
    hi = lo--;
#pragma clang loop unroll(disable)
    for (int j = start; j < end; j++) {
        int N = eB[j + 1] - eB[j];
        int bits1j = (C * N * CLT_band_allocation[lo * len + j]) << LM >> 2;
        int bits2j = hi >= 11 ? uni(F.cap[j]) : (C * N * CLT_band_allocation[hi * len + j]) << LM >> 2;
        if (bits1j > 0) bits1j = imax(0, bits1j + uni(F.trim_offset[j]));
        if (bits2j > 0) bits2j = imax(0, bits2j + uni(F.trim_offset[j]));
        if (lo > 0) bits1j += uni(F.offsets[j]);
        bits2j += uni(F.offsets[j]);
        if (uni(F.offsets[j]) > 0) skip_start = j;
        bits2j = imax(0, bits2j - bits1j);
        st0(&F.bits1[j], bits1j);
        st0(&F.bits2[j], bits2j);
    }
    wave_sync();
    // ---- interp_bits2pulses ----
    const int alloc_floor = C << BITRES, stereo = C > 1, logM = LM << BITRES;
    i32 psum;
    lo = 0;
    hi = 1 << ALLOC_STEPS;
    for (int i = 0; i < ALLOC_STEPS; i++) {
        int mid = (lo + hi) >> 1, done = 0;
        psum = 0;
        for (int j = end; j-- > start;) {
            int tmp = uni(F.bits1[j]) + ((mid * (i32)uni(F.bits2[j])) >> ALLOC_STEPS);
            if (tmp >= uni(F.thresh[j]) || done) { done = 1; psum += imin(tmp, uni(F.cap[j])); }
            else if (tmp >= alloc_floor) psum += alloc_floor;
        }
        if (psum > total) hi = mid; else lo = mid;
    }
    psum = 0;
    {
        int done = 0;
        for (int j = end; j-- > start;) {
            int tmp = uni(F.bits1[j]) + ((lo * uni(F.bits2[j])) >> ALLOC_STEPS);
            if (tmp < uni(F.thresh[j]) && !done) tmp = tmp >= alloc_floor ? alloc_floor : 0;
            else done = 1;
            tmp = imin(tmp, uni(F.cap[j]));
            st0(&bits[j], tmp);
            psum += tmp;
        }
    }
    wave_sync();
    int codedBands;
    for (codedBands = end;; codedBands--) {
        int j = codedBands - 1;
        if (j <= skip_start) { total += skip_rsv; break; }
        i32 left = total - psum;
        i32 percoeff = (u32)left / (u32)(eB[codedBands] - eB[start]);
        left -= (eB[codedBands] - eB[start]) * percoeff;
        int rem = imax(left - (eB[j] - eB[start]), 0);
        int band_width = eB[codedBands] - eB[j];
        int band_bits = (int)(uni(bits[j]) + percoeff * band_width + rem);
        if (band_bits >= imax(uni(F.thresh[j]), alloc_floor + (1 << BITRES))) {
            // encoder: keep the band if it is worth it and say so; decoder: read the decision (rate.c:352-376)
            if (coder_bit_logp(ec, codedBands <= start + 2 || (band_bits > (((j < prev ? 7 : 9) * band_width) << LM << BITRES) >> 4 && j <= signalBandwidth), 1))
                break;
            psum += 1 << BITRES;
            band_bits -= 1 << BITRES;
        }
        psum -= uni(bits[j]) + intensity_rsv;
        if (intensity_rsv > 0) intensity_rsv = CLT_log2_frac_table[j - start];
        psum += intensity_rsv;
        if (band_bits >= alloc_floor) { psum += alloc_floor; st0(&bits[j], (i32)alloc_floor); }
        else st0(&bits[j], 0);
    }
    if (intensity_rsv > 0) {
        out.intensity = imin(out.intensity, codedBands);
        out.intensity = start + (int)coder_uint(ec, (u32)(out.intensity - start), (u32)(codedBands + 1 - start));
    } else {
        out.intensity = 0;
    }
    if (out.intensity <= start) { total += dual_stereo_rsv; dual_stereo_rsv = 0; }
    if (dual_stereo_rsv > 0) out.dual_stereo = coder_bit_logp(ec, out.dual_stereo, 1);
    else out.dual_stereo = 0;
    wave_sync();
    i32 left = total - psum;
    i32 percoeff = (u32)left / (u32)(eB[codedBands] - eB[start]);
    left -= (eB[codedBands] - eB[start]) * percoeff;
    for (int j = start; j < codedBands; j++) st0(&bits[j], uni(bits[j]) + (int)percoeff * (eB[j + 1] - eB[j]));
    for (int j = start; j < codedBands; j++) {
        int tmp = (int)imin(left, eB[j + 1] - eB[j]);
        st0(&bits[j], uni(bits[j]) + tmp);
        left -= tmp;
    }
    i32 balance = 0;
    int j;
    for (j = start; j < codedBands; j++) {
        int N0 = eB[j + 1] - eB[j], N = N0 << LM;
        i32 bit = (i32)uni(bits[j]) + balance, excess;
        i32 bj, ej, fp;
        if (N > 1) {
            excess = imax(bit - uni(F.cap[j]), 0);
            bj = bit - excess;
            int den = C * N + ((C == 2 && N > 2 && !out.dual_stereo && j < out.intensity) ? 1 : 0);
            int NClogN = den * (CLT_logN400[j] + logM);
            int offset = (NClogN >> 1) - den * FINE_OFFSET;
            if (N == 2) offset += den << BITRES >> 2;
            if (bj + offset < (den * 2) << BITRES) offset += NClogN >> 2;
            else if (bj + offset < (den * 3) << BITRES) offset += NClogN >> 3;
            ej = imax(0, bj + offset + (den << (BITRES - 1)));
            ej = (i32)((u32)ej / (u32)den) >> BITRES;
            if (C * ej > (bj >> BITRES)) ej = bj >> stereo >> BITRES;
            ej = imin(ej, MAX_FINE_BITS);
            fp = ej * (den << BITRES) >= bj + offset;
            bj -= (C * ej) << BITRES;
        } else {
            excess = imax(0, bit - (C << BITRES));
            bj = bit - excess;
            ej = 0;
            fp = 1;
        }
        if (excess > 0) {
            int extra_fine = imin(excess >> (stereo + BITRES), MAX_FINE_BITS - ej);
            ej += extra_fine;
            int extra_bits = (extra_fine * C) << BITRES;
            fp = extra_bits >= excess - balance;
            excess -= extra_bits;
        }
        balance = excess;
        st0(&bits[j], bj);
        st0(&ebits[j], ej);
        st0(&fine_priority[j], fp);
    }
    out.balance = balance;
    for (; j < end; j++) {
        i32 e = uni(bits[j]) >> stereo >> BITRES;
        st0(&ebits[j], e);
        st0(&bits[j], 0);
        st0(&fine_priority[j], (i32)(e < 1));
    }
    wave_sync();
    out.codedBands = codedBands;
    return out;
}

template <class L>
CA_DEVFN void quant_fine_energy_wave(L &F, RangeEnc &enc, int C)                    // quant_bands.c:369-404
{
    for (int i = 0; i < NB; i++) {
        int fq = uni((i32)F.fine_quant[i]);
        if (fq <= 0) continue;
        i32 frac = (i16)(1 << fq);
        for (int c = 0; c < C; c++) {
            int q2 = (uni((i32)F.error[i + c * NB]) + 512) >> (10 - fq);
            if (q2 > frac - 1) q2 = frac - 1;
            if (q2 < 0) q2 = 0;
            ec_enc_bits(enc, (u32)q2, (u32)fq);
            i32 offset = (i16)sub16((shl32(q2, 10) + 512) >> fq, 512);
            st0(&F.oldBandE[i + c * NB], (i16)(uni((i32)F.oldBandE[i + c * NB]) + offset));
            st0(&F.error[i + c * NB], (i16)(uni((i32)F.error[i + c * NB]) - offset));
        }
    }
    wave_sync();
}

template <class L>
CA_DEVFN void quant_energy_finalise_wave(L &F, RangeEnc &enc, int bits_left, int C)   // quant_bands.c:406-439
{
    for (int prio = 0; prio < 2; prio++) {
        for (int i = 0; i < NB && bits_left >= C; i++) {
            if (uni((i32)F.fine_quant[i]) >= MAX_FINE_BITS || uni((i32)F.fine_priority[i]) != prio) continue;
            for (int c = 0; c < C; c++) {
                int q2 = uni((i32)F.error[i + c * NB]) < 0 ? 0 : 1;
                ec_enc_bits(enc, (u32)q2, 1);
                i32 offset = (i16)((shl16(q2, 10) - 512) >> (uni((i32)F.fine_quant[i]) + 1));
                st0(&F.oldBandE[i + c * NB], (i16)(uni((i32)F.oldBandE[i + c * NB]) + offset));
                bits_left--;
            }
        }
    }
    wave_sync();
}

// ---- PVQ ---------------------------------------------------------------------------------------------
// what quant_all_bands reads per band, wherever the build keeps it
#if defined(CA_LANE_FRAME)
template <class L> CA_DEV i32 frame_pulses(L &F, int i) { return lds_col(F.col)[LS_PULSES + i]; }
template <class L> CA_DEV int frame_tf_change(L &F, int i) { return (CLT_tf_select_table + LM3 * 8)[F.tf_sel + (int)((F.tf_bits >> i) & 1u)]; }
template <class L> CA_DEV i32 frame_bandE(L &F, int k) { return F.mid->bandE[k]; }
#else
template <class L> CA_DEV i32 frame_pulses(L &F, int i) { return uni(F.pulses[i]); }
template <class L> CA_DEV int frame_tf_change(L &F, int i) { return uni(F.tf_res[i]); }
template <class L> CA_DEV i32 frame_bandE(L &F, int k) { return uni(F.bandE[k]); }
#endif

struct BandCtx {                 // uniform; cf. struct band_ctx (bands.c:623-635)
    int i, intensity, spread, tf_change;
    i32 remaining_bits;
};

// exp_rotation1 chains (vq.c:43-68): positions r, r+stride, ... of one block form a serial recurrence; the
// `stride` residues and the blocks are independent, one lane walks one chain.
template <class P>
CA_DEV void exp_rotation1_chains(P X, int len, int nblocks, int stride, i32 c, i32 s)
{
    const i32 ms = (i16)neg32(s);
    if (LANES == 1 && stride == 1 && (nblocks & 3) == 0 && len >= 2) {
        // short blocks (B = 4 or 8 blocks of len = N / B bins, one chain each): four blocks in lockstep, so that a step is one
        // round trip of four loads for four chains instead of one per chain (a block of a short-block band is 1-6 steps long:
        // the look-ahead of the long-block path below never starts on it)
        for (int b0 = 0; b0 < nblocks; b0 += 4) {
            P x = X + b0 * len;
            i32 x1[4], x2[4];
#pragma unroll
            for (int g = 0; g < 4; g++) x1[g] = x[g * len];
            for (int i = 0; i < len - 1; i++) {
#pragma unroll
                for (int g = 0; g < 4; g++) x2[g] = x[g * len + i + 1];
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const i32 n2 = (i16)pshr32(mac16_16(mul16_16(c, x2[g]), s, x1[g]), 15);
                    x[g * len + i] = (i16)pshr32(mac16_16(mul16_16(c, x1[g]), ms, x2[g]), 15);
                    x1[g] = n2;
                }
            }
#pragma unroll
            for (int g = 0; g < 4; g++) x[g * len + len - 1] = (i16)x1[g];
            if (len >= 3) {
                // backward: i = len-3 .. 0; the value written to x[i] is the x2 of the next (lower) step
#pragma unroll
                for (int g = 0; g < 4; g++) x2[g] = x[g * len + len - 2];
                for (int i = len - 3; i >= 0; i--) {
#pragma unroll
                    for (int g = 0; g < 4; g++) x1[g] = x[g * len + i];
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        x[g * len + i + 1] = (i16)pshr32(mac16_16(mul16_16(c, x2[g]), s, x1[g]), 15);
                        x2[g] = (i16)pshr32(mac16_16(mul16_16(c, x1[g]), ms, x2[g]), 15);
                    }
                }
#pragma unroll
                for (int g = 0; g < 4; g++) x[g * len] = (i16)x2[g];
            }
        }
        wave_sync();
        return;
    }
    CA_UNROLL_LANE
    for (int ch = lane(); ch < nblocks * stride; ch += LANES) {
        P x = X + (ch / stride) * len;
        const int r = ch % stride;
        // forward: the value written to x[i+stride] is the x1 of the next step -> carried in a register,
        // so each step needs one independent LDS read instead of a read-after-write round trip
        if (r < len - stride) {
            i32 x1 = x[r];
            int i = r;
            if (LANES == 1) {
                // four look-ahead loads before the first store of the group (they never alias: the stores
                // trail the loads by `stride`). (Software pipelines -- the NEXT group loaded before this one is computed, here
                // and in the greedy search of alg_quant_lane -- were slower at 64 frames per wavefront, 3.21 -> 3.26 / 3.29 ms: the
                // compiler already waits per element, lgkmcnt(n), and the copies of the look-ahead registers cost more.)
                for (; i + 3 * stride < len - stride; i += 4 * stride) {
                    i32 v[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) v[u] = x[i + (u + 1) * stride];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        i32 n2 = (i16)pshr32(mac16_16(mul16_16(c, v[u]), s, x1), 15);
                        x[i + u * stride] = (i16)pshr32(mac16_16(mul16_16(c, x1), ms, v[u]), 15);
                        x1 = n2;
                    }
                }
            }
            for (; i < len - stride; i += stride) {
                i32 x2 = x[i + stride];
                i32 n2 = (i16)pshr32(mac16_16(mul16_16(c, x2), s, x1), 15);
                x[i] = (i16)pshr32(mac16_16(mul16_16(c, x1), ms, x2), 15);
                x1 = n2;
            }
            x[i] = (i16)x1;
        }
        // backward pass: i = len-2*stride-1 .. 0 restricted to this residue; the value written to x[i]
        // is the x2 of the next (lower) step
        int top = len - 2 * stride - 1;
        if (top >= r) {
            int i = top - ((top - r) % stride);
            i32 x2 = x[i + stride];
            if (LANES == 1) {
                for (; i - 3 * stride >= 0; i -= 4 * stride) {
                    i32 v[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) v[u] = x[i - u * stride];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        x[i - u * stride + stride] = (i16)pshr32(mac16_16(mul16_16(c, x2), s, v[u]), 15);
                        x2 = (i16)pshr32(mac16_16(mul16_16(c, v[u]), ms, x2), 15);
                    }
                }
            }
            for (; i >= 0; i -= stride) {
                i32 x1 = x[i];
                x[i + stride] = (i16)pshr32(mac16_16(mul16_16(c, x2), s, x1), 15);
                x2 = (i16)pshr32(mac16_16(mul16_16(c, x1), ms, x2), 15);
            }
            x[i + stride] = (i16)x2;
        }
    }
    wave_sync();
}

template <class P>
CA_DEV void exp_rotation_wave(P X, int len, int stride, int K, int spread)              // vq.c:70-117, dir = 1
{
    if (2 * K >= len || spread == SPREAD_NONE) return;
    const int factor = spread == SPREAD_LIGHT ? 15 : spread == SPREAD_NORMAL ? 10 : 5;   // SPREAD_FACTOR[spread-1]
    i32 gain = (i16)celt_div(mul16_16(32767, len), len + factor * K);
    i32 theta = (i16)mul16_16_q15(gain, gain) >> 1;
    i32 c = celt_cos_norm(theta);
    i32 s = celt_cos_norm((i16)sub16(32767, theta));
    int stride2 = 0;
    if (len >= 8 * stride) {
        stride2 = 1;
        while ((stride2 * stride2 + stride2) * stride + (stride >> 2) < len) stride2++;
    }
    const int blen = (int)((u32)len / (u32)stride);
    exp_rotation1_chains(X, blen, stride, 1, c, (i16)neg32(s));
    if (stride2) exp_rotation1_chains(X, blen, stride, stride2, s, (i16)neg32(c));
}

// encode_pulses(iy, N, K) = ec_enc_uint(icwrs(N, iy), V(N,K))  (cwrs.c:440-460)
template <class L, class PI>
CA_DEVFN void encode_pulses_wave(L &F, RangeEnc &ec, int N, int K, PI y)
{
    u32 idx;
    if constexpr (LANES == 1) {
        // one lane owns the frame: icwrs as the reference walks it (cwrs.c:440-456), no suffix-sum array
        int j = N - 1;
        idx = (u32)(y[j] < 0);
        int k = y[j] < 0 ? -y[j] : y[j];
        do {
            j--;
            idx += pvq_u(N - j, k);
            k += y[j] < 0 ? -y[j] : y[j];
            if (y[j] < 0) idx += pvq_u(N - j, k + 1);
        } while (j > 0);
    } else {
    // suffix sums k_j = sum_{t>=j} |y_t| are produced serially (cheap), the table look-ups in parallel
    i16 *suf = F.s.pvq.xabs;                         // free after the search: reuse as suffix sums
    if (lane() == 0) {
        int k = 0;
        for (int j = N - 1; j >= 0; j--) { int a = y[j]; k += a < 0 ? -a : a; suf[j] = (i16)k; }
    }
    wave_sync();
    u32 p = 0;
    CA_UNROLL_LANE
    for (int j = lane(); j < N - 1; j += LANES) {
        p += pvq_u(N - j, suf[j + 1]);
        if (y[j] < 0) p += pvq_u(N - j, suf[j] + 1);
    }
    idx = (u32)wave_add((i32)p) + (u32)(uni(y[N - 1]) < 0);
    }
    u32 V = pvq_u(N, K) + pvq_u(N, K + 1);
    wave_sync();
    ec_enc_uint(ec, idx, V);
}

// Wave arg-max of (num/den) with index tie-break. `o` beats `m` iff m.den*o.num > o.den*m.num, or the
// cross products are equal and o.id < m.id -- the order the sequential scan of vq.c:277-300 induces.
#if defined(CA_SINGLE_LANE)
CA_DEV void pvq_argmax(i32 &, i32 &, int &) {}
#else
#define CA_ARGMAX_STEP(ctrl, rowmask)                                                          \
    do {                                                                                        \
        i32 on = CA_DPP(-32767, num, ctrl, rowmask), od = CA_DPP(0, den, ctrl, rowmask);        \
        int oi = CA_DPP(0x7fff, id, ctrl, rowmask);                                             \
        i32 lhs = __mul24(den, on), rhs = __mul24(od, num);                                     \
        bool take = lhs > rhs || (lhs == rhs && oi < id);                                       \
        num = take ? on : num; den = take ? od : den; id = take ? oi : id;                      \
    } while (0)
CA_DEV void pvq_argmax(i32 &num, i32 &den, int &id)
{
    CA_ARGMAX_STEP(0x111, 0xf);
    CA_ARGMAX_STEP(0x112, 0xf);
    CA_ARGMAX_STEP(0x114, 0xf);
    CA_ARGMAX_STEP(0x118, 0xf);
    CA_ARGMAX_STEP(0x142, 0xa);
    CA_ARGMAX_STEP(0x143, 0xc);
    id = __builtin_amdgcn_readlane(id, 63);
}
#endif

#if defined(CA_LANE_FRAME)
// ---- lane build: the leaf quantiser on this lane's LDS columns, written for the instruction stream it produces --------
// One lane owns the leaf, 32 (or 64) leaves of different N and K share the wavefront. What costs here is (a) every
// instruction of the wavefront's serial timeline, address arithmetic included, and (b) the trip counts of nested loops,
// which a wavefront pays as max x max over its lanes. Hence:
//  * columns are walked in chunks of eight elements through ONE moving pointer, so that the eight accesses of a chunk
//    are ds_read / ds_write with immediate offsets (u * 128 bytes) instead of eight address computations;
//  * the greedy search runs as ONE loop over (pulse, chunk) steps per lane: the wavefront pays max(pulses x chunks)
//    instead of max(pulses) x max(chunks);
//  * 2*iy is not stored (Ryy = yy + 2*iy[j] is one v_lshl_add), the signs of X are kept in a 64-bit mask and |X|
//    overwrites the leaf copy in place;
//  * icwrs runs in two passes per chunk -- the running pulse counts in registers, then all table look-ups of the chunk
//    issued together -- instead of three dependent LDS round trips per element.
// Arithmetic and the order of every comparison are those of vq.c:161-325 / cwrs.c:440-460.
enum { LDS_COL = 64 };          // elements between consecutive entries of one lane's column

// (i16)a * (i16)b as ONE full-rate instruction: the 24-bit multiplier on the sign-extended low halves (SDWA selects)
CA_DEV i32 mul16x16_lo(i32 a, i32 b)
{
#if defined(CA_HOST_EMU)
    return (i32)(i16)a * (i32)(i16)b;
#endif
    i32 r;
    asm("v_mul_i32_i24_sdwa %0, sext(%1), sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_0"
        : "=v"(r) : "v"(a), "v"(b));
    return r;
}

CA_DEV u32 pvq_u_lds(int n, int k)
{
    const int lo = n < k ? n : k, hi = n < k ? k : n;
    return CLT_pvq_u_data[CLT_pvq_u_row[lo] + hi];
}

// One lane's LDS column, in 16-bit slots (celt_lane_tables.h: 240 slots per lane): two band buffers of LANE_HALF bins
// (bands of up to 96 bins ping-pong between them for the de-interleave; the two widest bands, 144 and 176 bins, lie
// across both) and LANE_IYN slots of pulse counts for leaves of up to 48 bins.
enum { LANE_HALF = 96, LANE_IY = 192, LANE_IYN = 48 };

// X: the leaf, in place in the band buffer (becomes |X|); iy: N slots (rounded up to a multiple of eight) for the pulse
// counts, bit 15 of a slot = "X[j] <= 0" (signx of vq.c:189-199; the counts stay below 2^8, and 2 * slot read as a 16-bit
// number is 2 * count whatever bit 15 holds).
template <class L>
CA_DEVFN void alg_quant_lane(L &F, RangeEnc &ec, CA_AS_LDS i16 *const X, CA_AS_LDS u16 *const iy, int N, int K, int spread, int B)
{
    CA_STAMP_F(F, 22);
    exp_rotation_wave(lds_col(X), N, B, K, spread);
    CA_STAMP_F(F, 17);
    const int nch = (N + 7) >> 3;
    // |X| in place, iy = sign flag, sum |X|; slots past N (up to the next multiple of eight) are zeroed
    i32 sum = 0;
    {
        CA_AS_LDS i16 *qx = X;
        CA_AS_LDS u16 *qi = iy;
        for (int c = 0; c < nch; c++, qx += 8 * LDS_COL, qi += 8 * LDS_COL) {
            i32 v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = qx[u * LDS_COL];
            const int rem = N - c * 8;
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const bool in = u < rem;
                const i32 a = in ? (i32)(i16)(v[u] > 0 ? v[u] : -v[u]) : 0;
                sum += a;
                qx[u * LDS_COL] = (i16)a;
                qi[u * LDS_COL] = (u16)((in & (v[u] <= 0)) ? 0x8000 : 0);
            }
        }
    }
    i32 xy = 0, yy = 0;
    int pulsesLeft = K;
    if (K > (N >> 1)) {
        if (sum <= K) {
            CA_AS_LDS i16 *qx = X;
            for (int c = 0; c < nch; c++, qx += 8 * LDS_COL)
#pragma unroll
                for (int u = 0; u < 8; u++) qx[u * LDS_COL] = 0;
            X[0] = 16384;
            sum = 16384;
        }
        const i32 rcp = (i16)mul16_32_q16((i16)(K - 1), celt_rcp(sum));
        i32 pyy = 0, ppl = 0;
        CA_AS_LDS i16 *qx = X;
        CA_AS_LDS u16 *qi = iy;
        for (int c = 0; c < nch; c++, qx += 8 * LDS_COL, qi += 8 * LDS_COL) {
            i32 a[8], w[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { a[u] = qx[u * LDS_COL]; w[u] = qi[u * LDS_COL]; }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const i32 q = mul16_16_q15(a[u], rcp);           // pad slots hold 0 -> q = 0
                qi[u * LDS_COL] = (u16)(w[u] | q);
                const i32 yj = (i16)q;
                pyy = mac16_16(pyy, yj, yj);
                xy = mac16_16(xy, a[u], yj);
                ppl += q;
            }
        }
        yy = (i16)pyy;
        pulsesLeft -= ppl;
    }
    if (pulsesLeft > N + 3) {
        const i32 tmp = (i16)pulsesLeft;
        const i32 w0 = iy[0];
        const i32 y0 = (i16)(2 * w0);
        yy = (i16)mac16_16(yy, tmp, tmp);
        yy = (i16)mac16_16(yy, tmp, y0);
        iy[0] = (u16)(w0 + pulsesLeft);
        pulsesLeft = 0;
    }
    CA_STAMP_F(F, 18);
    // greedy search (vq.c:259-306): steps = pulses x chunks, one chunk of eight positions per step
    {
        const int steps = pulsesLeft * nch;
        int i = 0, c = 0, rshift = 0;
        i32 best_num = 0, best_den = 0;
        int best_id = 0;
        CA_AS_LDS i16 *qx = X;
        CA_AS_LDS u16 *qi = iy;
        for (int s = 0; s < steps; s++) {
            if (c == 0) {
                rshift = 1 + celt_ilog2(K - pulsesLeft + i + 1);
                yy = (i16)add32(yy, 1);
                best_num = -32767;
                best_den = 0;
                best_id = 0;
                qx = X;
                qi = iy;
            }
            i32 a[8], w[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { a[u] = qx[u * LDS_COL]; w[u] = qi[u * LDS_COL]; }
            int best_u = -1;                            // position inside this chunk of the best candidate so far, if it is here
            const int rem = N - c * 8;                  // positions u < rem of this chunk exist
#pragma unroll
            for (int u = 0; u < 8; u++) {
                // Rxy, Ryy, best_num, best_den are 16-bit quantities of the reference (opus_val16); they are kept here
                // WITHOUT the truncating sign extension and every consumer reads their low half sign-extended (SDWA), which
                // is the same number. Both cross products are 16 x 16 bit: the full-rate 24-bit multiplier, stated as such --
                // left to itself the compiler loses the operand ranges through the loop-carried best_* and emits quarter-rate
                // v_mul_lo_u32. No short-circuit either: a branch per position costs more than the multiplies it would skip.
                const i32 t = add32(xy, a[u]) >> rshift;
                const i32 Rxy = mul16x16_lo(t, t) >> 15;
                const i32 Ryy = yy + 2 * w[u];          // low half: yy + 2 * count (the sign flag shifts out of it)
                const bool take = (u < rem) & (mul16x16_lo(best_den, Rxy) > mul16x16_lo(Ryy, best_num));
                best_den = take ? Ryy : best_den;
                best_num = take ? Rxy : best_num;
                best_u = take ? u : best_u;
            }
            best_id = best_u >= 0 ? c * 8 + best_u : best_id;
            qx += 8 * LDS_COL;
            qi += 8 * LDS_COL;
            if (++c == nch) {
                const i32 wb = iy[best_id * LDS_COL];
                xy = add32(xy, (i32)X[best_id * LDS_COL]);
                yy = add16(yy, (i16)(2 * wb));
                iy[best_id * LDS_COL] = (u16)(wb + 1);
                c = 0;
                i++;
            }
        }
    }
    CA_STAMP_F(F, 19);
    // encode_pulses: icwrs(N, y), y[j] = flag[j] ? -count[j] : count[j]  (cwrs.c:440-460), walked from the last element down
    u32 idx;
    {
        int j = N - 1;
        const i32 wl = iy[j * LDS_COL];
        i32 k = wl & 0x7fff;
        idx = (u32)(wl > 0x8000);
        // element e contributes U(N-e, pulses above it) and, if negative, U(N-e, pulses above it + its own + 1).
        // Straight-line on purpose: every load of a chunk is unconditional (indices clamped into range, results masked
        // afterwards), so that the eight count loads, then the sixteen row look-ups, then the sixteen table look-ups are
        // each issued back to back and waited for once -- a conditional look-up becomes a branch with its own LDS round trip.
        while (j > 0) {
            const int j0 = j - 1;                      // highest element of this chunk
            const int n8 = j0 + 1 < 8 ? j0 + 1 : 8;    // elements j0, j0-1, ..., j0-n8+1
            i32 wv[8];
#pragma unroll
            for (int u = 0; u < 8; u++) wv[u] = (i32)iy[(j0 - u > 0 ? j0 - u : 0) * LDS_COL];
            int k1[8], k2[8], nn[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const bool in = u < n8;
                const i32 m = in ? (wv[u] & 0x7fff) : 0;
                nn[u] = in ? N - (j0 - u) : 1;
                k1[u] = in ? k : 0;
                k += m;
                k2[u] = in ? k + 1 : 0;
            }
            int row1[8], row2[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                row1[u] = CLT_pvq_u_row[nn[u] < k1[u] ? nn[u] : k1[u]];
                row2[u] = CLT_pvq_u_row[nn[u] < k2[u] ? nn[u] : k2[u]];
            }
            u32 r1[8], r2[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                r1[u] = CLT_pvq_u_data[imin(row1[u] + (nn[u] < k1[u] ? k1[u] : nn[u]), 1271)];
                r2[u] = CLT_pvq_u_data[imin(row2[u] + (nn[u] < k2[u] ? k2[u] : nn[u]), 1271)];
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const bool in = u < n8;
                idx += (in ? r1[u] : 0u) + ((in & (wv[u] > 0x8000)) ? r2[u] : 0u);
            }
            j -= n8;
        }
    }
    const u32 V = pvq_u_lds(N, K) + pvq_u_lds(N, K + 1);
    ec_enc_uint(ec, idx, V);
    CA_STAMP_F(F, 20);
}

#endif

// alg_quant(X, N, K, spread, B, enc)  (vq.c:161-325), non-RESYNTH build.
// The body is a template over where the leaf X and the search state (2*iy, |X|, iy) live: plain pointers into the
// wave's LDS struct (wave build, host emulation); in the lane build either this lane's columns of the workgroup's LDS
// scratch (leaves of up to PVQ_LDS_N bins: typed LdsCol accessors -> ds_read / ds_write) or, for the rare larger leaf,
// private memory and the frame's own row in HBM.
template <class L, class PX, class PY, class PI>
CA_DEVFN void alg_quant_body(L &F, RangeEnc &ec, PX X, PY y, PY xa, PI iy, int N, int K, int spread, int B);

template <class L>
CA_DEVFN void alg_quant_wave(L &F, RangeEnc &ec, x16_t *Xg, int N, int K, int spread, int B)
{
#if defined(CA_LANE_FRAME)
    if (N <= PVQ_LDS_N) {
        // leaves of up to PVQ_LDS_N bins are rotated, searched and sign-fixed on an LDS copy ([element][lane]); the
        // encoder never reads X again after coding it, so nothing is copied back
        LdsCol<i16> st = lds_col(F.lds_xs);
        if (((uintptr_t)Xg & 15) == 0) {
            // 16-byte aligned leaf: whole groups of eight bins per load (a group may reach past the leaf into the
            // following bins of the same frame; they are loaded, not used)
            for (int k = 0; k < N; k += 8) {
                i32 v[8];
                ld_bins8(Xg + k, v);
#pragma unroll
                for (int u = 0; u < 8; u++)
                    if (k + u < N) st[k + u] = (i16)v[u];
            }
        } else {
            for (int k = 0; k < N; k += 8) {
                i32 v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = k + u < N ? (i32)Xg[k + u] : 0;
#pragma unroll
                for (int u = 0; u < 8; u++)
                    if (k + u < N) st[k + u] = (i16)v[u];
            }
        }
        alg_quant_lane(F, ec, F.lds_xs, (CA_AS_LDS u16 *)F.lds_pvq16, N, K, spread, B);
    } else {
        alg_quant_body(F, ec, Xg, priv((i16 *)F.s.pvq.y), priv((i16 *)F.s.pvq.xabs), priv((i32 *)F.s.pvq.iy), N, K, spread, B);
    }
#else
    alg_quant_body(F, ec, Xg, (i16 *)F.s.pvq.y, (i16 *)F.s.pvq.xabs, (i32 *)F.s.pvq.iy, N, K, spread, B);
#endif
}

template <class L, class PX, class PY, class PI>
CA_DEVFN void alg_quant_body(L &F, RangeEnc &ec, PX X, PY y, PY xa, PI iy, int N, int K, int spread, int B)
{
    CA_STAMP_F(F, 22);
    CA_COUNT("leaf.N", N);
    CA_COUNT("leaf.K", K);
    CA_COUNT(2 * K >= N || spread == SPREAD_NONE ? "leaf.norot" : "leaf.rot", N);
    CA_COUNT(N <= 16 ? "leaf.N<=16" : N <= 32 ? "leaf.N<=32" : N <= 64 ? "leaf.N<=64" : "leaf.N>64", N);
    exp_rotation_wave(X, N, B, K, spread);
    CA_STAMP_F(F, 17);
    CA_UNROLL_LANE
    for (int j = lane(); j < N; j += LANES) {
        i32 v = X[j];
        xa[j] = (i16)(v > 0 ? v : -v);
        iy[j] = 0;
        y[j] = 0;
    }
    wave_sync();
    i32 xy = 0, yy = 0;
    int pulsesLeft = K;
    if (K > (N >> 1)) {
        i32 p = 0;
        CA_UNROLL_LANE
        for (int j = lane(); j < N; j += LANES) p += xa[j];
        i32 sum = wave_add(p);
        if (sum <= K) {
            CA_UNROLL_LANE
            for (int j = lane(); j < N; j += LANES) xa[j] = j == 0 ? 16384 : 0;
            sum = 16384;
            wave_sync();
        }
        i32 rcp = (i16)mul16_32_q16((i16)(K - 1), celt_rcp(sum));
        i32 pyy = 0, pxy = 0, ppl = 0;
        CA_UNROLL_LANE
        for (int j = lane(); j < N; j += LANES) {
            i32 q = mul16_16_q15(xa[j], rcp);
            iy[j] = q;
            i32 yj = (i16)q;
            pyy = mac16_16(pyy, yj, yj);
            pxy = mac16_16(pxy, xa[j], yj);
            y[j] = (i16)(yj * 2);
            ppl += q;
        }
        yy = (i16)wave_add(pyy);
        xy = wave_add(pxy);
        pulsesLeft -= wave_add(ppl);
        wave_sync();
    }
    if (pulsesLeft > N + 3) {
        i32 tmp = (i16)pulsesLeft;
        yy = (i16)mac16_16(yy, tmp, tmp);
        yy = (i16)mac16_16(yy, tmp, uni((i32)y[0]));
        st0(&iy[0], uni(iy[0]) + pulsesLeft);
        pulsesLeft = 0;
        wave_sync();
    }
    CA_STAMP_F(F, 18);
    CA_COUNT(K > (N >> 1) ? "leaf.presearch" : "leaf.nopresearch", pulsesLeft);
    CA_COUNT("greedy.pulses", pulsesLeft);
    CA_COUNT("greedy.pulses*chunks", pulsesLeft * ((N + 63) / 64));
    for (int i = 0; i < pulsesLeft; i++) {
        const int rshift = 1 + celt_ilog2(K - pulsesLeft + i + 1);
        yy = (i16)add32(yy, 1);
        i32 best_num = -32767, best_den = 0;
        int best_id = 0;
        CA_UNROLL_LANE
        for (int j = lane(); j < N; j += LANES) {
            i32 Rxy = (i16)(add32(xy, xa[j]) >> rshift);
            i32 Ryy = add16(yy, y[j]);
            Rxy = (i16)mul16_16_q15(Rxy, Rxy);
            if (mul16_16(best_den, Rxy) > mul16_16(Ryy, best_num)) { best_den = Ryy; best_num = Rxy; best_id = j; }
        }
        // wave arg-max under the same (exact, cross-multiplied) order; ties go to the lower index
        pvq_argmax(best_num, best_den, best_id);
        xy = add32(xy, uni((i32)xa[best_id]));
        yy = add16(yy, uni((i32)y[best_id]));
        if (lane() == (best_id & (LANES - 1))) { y[best_id] = (i16)(y[best_id] + 2); iy[best_id] = iy[best_id] + 1; }
        wave_sync();
    }
    CA_STAMP_F(F, 19);
    CA_UNROLL_LANE
    for (int j = lane(); j < N; j += LANES)
        if (X[j] <= 0) iy[j] = -iy[j];                                            // signx[j] < 0  <=>  X[j] <= 0
    wave_sync();
    encode_pulses_wave(F, ec, N, K, iy);
    CA_STAMP_F(F, 20);
}

CA_DEV int stereo_itheta_wave(const x16_t *X, const x16_t *Y, int stereo, int N)               // vq.c:376-408
{
    i32 pm = 0, ps = 0;
#if defined(CA_LANE_FRAME)
    if ((N & 7) == 0 && (((uintptr_t)X | (uintptr_t)Y) & 15) == 0) {
#pragma unroll 4
        for (int i = 0; i < N; i += 8) {
            i32 xv[8], yv[8];
            ld_bins8(X + i, xv);
            ld_bins8(Y + i, yv);
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (stereo) {
                    i32 m = add16(xv[u] >> 1, yv[u] >> 1), sd = (i16)sub16(xv[u] >> 1, yv[u] >> 1);
                    pm = mac16_16(pm, m, m);
                    ps = mac16_16(ps, sd, sd);
                } else {
                    pm = mac16_16(pm, xv[u], xv[u]);
                    ps = mac16_16(ps, yv[u], yv[u]);
                }
            }
        }
    } else
#endif
    if (stereo) {
        CA_UNROLL_LANE
        for (int i = lane(); i < N; i += LANES) {
            i32 m = add16(X[i] >> 1, Y[i] >> 1), s = (i16)sub16(X[i] >> 1, Y[i] >> 1);
            pm = mac16_16(pm, m, m);
            ps = mac16_16(ps, s, s);
        }
    } else {
        CA_UNROLL_LANE
        for (int i = lane(); i < N; i += LANES) { pm = mac16_16(pm, X[i], X[i]); ps = mac16_16(ps, Y[i], Y[i]); }
    }
    i32 Emid = add32(1, wave_add(pm)), Eside = add32(1, wave_add(ps));
    i32 mid = (i16)celt_sqrt(Emid), side = (i16)celt_sqrt(Eside);
    return mul16_16_q15(20861, celt_atan2p(side, mid));                                   // QCONST16(0.63662f,15)
}

CA_DEV int compute_qn(int N, int b, int offset, int pulse_cap, int stereo)                 // bands.c:596-621
{
    int N2 = 2 * N - 1;
    if (stereo && N == 2) N2--;
    int qb = (b + N2 * offset) / N2;
    qb = imin(b - pulse_cap - (4 << BITRES), qb);
    qb = imin(8 << BITRES, qb);
    if (qb < (1 << BITRES >> 1)) return 1;
    int qn = CLT_exp2_table8[qb & 0x7] >> (14 - (qb >> BITRES));
    return (qn + 1) >> 1 << 1;
}

struct SplitCtx { int inv, imid, iside, delta, itheta, qalloc, staged; };   // staged (lane build): the child coded first is already in the column

// imid, iside and delta of a quantised angle (bands.c:789-811)
CA_DEV void theta_params(int itheta, int N, int *imid, int *iside, int *delta)
{
    if (itheta == 0) { *imid = 32767; *iside = 0; *delta = -16384; }
    else if (itheta == 16384) { *imid = 0; *iside = 32767; *delta = 16384; }
    else {
        *imid = bitexact_cos((i16)itheta);
        *iside = bitexact_cos((i16)(16384 - itheta));
        *delta = frac_mul16((N - 1) << 7, bitexact_log2tan(*iside, *imid));
    }
}

// compute_theta (bands.c:645-817), encode = 1
template <class L>
CA_DEVFN SplitCtx compute_theta_wave(L &F, RangeEnc &ec, BandCtx &ctx, x16_t *X, x16_t *Y, int N, int *b, int B,
                                     int B0, int LM, int stereo, int stage_first = 0)
{
    (void)B;
    (void)stage_first;
    SplitCtx sc;
    sc.staged = 0;
    int imid = 0, iside = 0, delta = 0;
    bool have_params = false;
    const int i = ctx.i;
    int inv = 0;
    CA_STAMP_F(F, 22);
    CA_COUNT(stereo ? "theta.stereo" : "theta.split", N);
    int pulse_cap = CLT_logN400[i] + LM * (1 << BITRES);
    int offset = (pulse_cap >> 1) - (stereo && N == 2 ? QTHETA_OFFSET_TWOPHASE : QTHETA_OFFSET);
    int qn = compute_qn(N, *b, offset, pulse_cap, stereo);
    if (stereo && i >= ctx.intensity) qn = 1;
    int itheta = stereo_itheta_wave(X, Y, stereo, N);
#if defined(CA_LANE_FRAME)
    const bool vec8 = (N & 7) == 0 && (((uintptr_t)X | (uintptr_t)Y) & 15) == 0;      // whole 16-byte groups of bins
#endif
    i32 tell = (i32)ec_tell_frac(ec);
    if (qn != 1) {
        itheta = (itheta * qn + 8192) >> 14;
        if (stereo && N > 2) {
            int p0 = 3, x = itheta, x0 = qn / 2, ft = p0 * (x0 + 1) + x0;
            ec_encode(ec, (u32)(x <= x0 ? p0 * x : (x - 1 - x0) + (x0 + 1) * p0),
                      (u32)(x <= x0 ? p0 * (x + 1) : (x - x0) + (x0 + 1) * p0), (u32)ft);
        } else if (B0 > 1 || stereo) {
            ec_enc_uint(ec, (u32)itheta, (u32)(qn + 1));
        } else {
            int ft = ((qn >> 1) + 1) * ((qn >> 1) + 1);
            int fs = itheta <= (qn >> 1) ? itheta + 1 : qn + 1 - itheta;
            int fl = itheta <= (qn >> 1) ? (itheta * (itheta + 1)) >> 1 : ft - (((qn + 1 - itheta) * (qn + 2 - itheta)) >> 1);
            ec_encode(ec, (u32)fl, (u32)(fl + fs), (u32)ft);
        }
        itheta = (int)((u32)(itheta * 16384) / (u32)qn);
        if (stereo) {
            if (itheta == 0) {
                // intensity_stereo (bands.c:336-360)
                i32 bl = frame_bandE(F, i), br = frame_bandE(F, i + NB);
                int shift = celt_zlog2(imax(bl, br)) - 13;
                i32 left = (i16)vshr32(bl, shift), right = (i16)vshr32(br, shift);
                i32 norm = (i16)(1 + celt_sqrt(add32(1, add32(mul16_16(left, left), mul16_16(right, right)))));
                i32 a1 = (i16)(shl32(left, 14) / norm), a2 = (i16)(shl32(right, 14) / norm);
#if defined(CA_LANE_FRAME)
                if (vec8)
                    for (int j = 0; j < N; j += 8) {
                        i32 xv[8], yv[8];
                        ld_bins8(X + j, xv);
                        ld_bins8(Y + j, yv);
#pragma unroll
                        for (int u = 0; u < 8; u++) xv[u] = (i16)(mac16_16(mul16_16(a1, xv[u]), a2, yv[u]) >> 14);
                        st_bins8(X + j, xv);
                    }
                else
#endif
                CA_UNROLL_LANE
                for (int j = lane(); j < N; j += LANES)
                    X[j] = (i16)(mac16_16(mul16_16(a1, X[j]), a2, Y[j]) >> 14);
            } else {
#if defined(CA_LANE_FRAME)
                if (vec8) {
                    // stereo_split (bands.c:362-373); two groups of eight bins per trip, all four loads ahead of the first store (loads
                    // queue behind stores: a group per trip is an exposed memory round trip per eight bins).
                    // The child that is coded first (quant_band_stereo orders them by their budgets, bands.c:1268-1275: the
                    // arithmetic below is the caller's) goes straight into this lane's column, where quant_band_lane works on
                    // it, and not back to its row: written there it would be read again at once, behind its own stores.
                    bool to_col = false, mid_first = true;
                    if (stage_first) {
                        theta_params(itheta, N, &imid, &iside, &delta);
                        have_params = true;
                        const i32 b1 = *b - (int)((i32)ec_tell_frac(ec) - tell);
                        const i32 mbits = imax(0, imin(b1, (b1 - delta) / 2));
                        mid_first = mbits >= b1 - mbits;
                        to_col = true;
                        sc.staged = 1;
                    }
                    CA_AS_LDS i16 *const col = F.col;
                    int j = 0;
                    for (; j + 16 <= N; j += 16) {
                        i32 xv[2][8], yv[2][8];
#pragma unroll
                        for (int g = 0; g < 2; g++) { ld_bins8(X + j + 8 * g, xv[g]); ld_bins8(Y + j + 8 * g, yv[g]); }
#pragma unroll
                        for (int g = 0; g < 2; g++) {
#pragma unroll
                            for (int u = 0; u < 8; u++) {
                                const i32 l = mul16_16(23170, xv[g][u]), r = mul16_16(23170, yv[g][u]);
                                xv[g][u] = (i16)(add32(l, r) >> 15);
                                yv[g][u] = (i16)(sub32(r, l) >> 15);
                            }
                            if (to_col) {
                                CA_AS_LDS i16 *c = col + (j + 8 * g) * LDS_COL;
#pragma unroll
                                for (int u = 0; u < 8; u++) c[u * LDS_COL] = (i16)(mid_first ? xv[g][u] : yv[g][u]);
                            }
                            if (!(to_col && mid_first)) st_bins8(X + j + 8 * g, xv[g]);
                            if (!(to_col && !mid_first)) st_bins8(Y + j + 8 * g, yv[g]);
                        }
                    }
                    for (; j < N; j += 8) {
                        i32 xv[8], yv[8];
                        ld_bins8(X + j, xv);
                        ld_bins8(Y + j, yv);
#pragma unroll
                        for (int u = 0; u < 8; u++) {
                            const i32 l = mul16_16(23170, xv[u]), r = mul16_16(23170, yv[u]);
                            xv[u] = (i16)(add32(l, r) >> 15);
                            yv[u] = (i16)(sub32(r, l) >> 15);
                        }
                        if (to_col) {
                            CA_AS_LDS i16 *c = col + j * LDS_COL;
#pragma unroll
                            for (int u = 0; u < 8; u++) c[u * LDS_COL] = (i16)(mid_first ? xv[u] : yv[u]);
                        }
                        if (!(to_col && mid_first)) st_bins8(X + j, xv);
                        if (!(to_col && !mid_first)) st_bins8(Y + j, yv);
                    }
                } else
#endif
                CA_UNROLL_LANE
                for (int j = lane(); j < N; j += LANES) {                                  // stereo_split (bands.c:362-373)
                    i32 l = mul16_16(23170, X[j]), r = mul16_16(23170, Y[j]);
                    X[j] = (i16)(add32(l, r) >> 15);
                    Y[j] = (i16)(sub32(r, l) >> 15);
                }
            }
            wave_sync();
        }
    } else if (stereo) {
        inv = itheta > 8192;
        if (inv)
            CA_UNROLL_LANE
            for (int j = lane(); j < N; j += LANES) Y[j] = (i16)(-Y[j]);
        wave_sync();
        {
            i32 bl = frame_bandE(F, i), br = frame_bandE(F, i + NB);
            int shift = celt_zlog2(imax(bl, br)) - 13;
            i32 left = (i16)vshr32(bl, shift), right = (i16)vshr32(br, shift);
            i32 norm = (i16)(1 + celt_sqrt(add32(1, add32(mul16_16(left, left), mul16_16(right, right)))));
            i32 a1 = (i16)(shl32(left, 14) / norm), a2 = (i16)(shl32(right, 14) / norm);
            CA_UNROLL_LANE
            for (int j = lane(); j < N; j += LANES)
                X[j] = (i16)(mac16_16(mul16_16(a1, X[j]), a2, Y[j]) >> 14);
        }
        wave_sync();
        if (*b > 2 << BITRES && ctx.remaining_bits > 2 << BITRES) ec_enc_bit_logp(ec, inv, 2);
        else inv = 0;
        itheta = 0;
    }
    int qalloc = (int)((i32)ec_tell_frac(ec) - tell);
    *b -= qalloc;
    if (!have_params) theta_params(itheta, N, &imid, &iside, &delta);
    CA_STAMP_F(F, 16);
    sc.inv = inv; sc.imid = imid; sc.iside = iside; sc.delta = delta; sc.itheta = itheta; sc.qalloc = qalloc;
    return sc;
}

CA_DEV void quant_band_n1_wave(RangeEnc &ec, BandCtx &ctx, const x16_t *X, const x16_t *Y)     // bands.c:819-862
{
    const x16_t *x = X;
    for (int c = 0; c < (Y ? 2 : 1); c++) {
        if (ctx.remaining_bits >= 1 << BITRES) {
            ec_enc_bits(ec, (u32)(uni((i32)x[0]) < 0), 1);
            ctx.remaining_bits -= 1 << BITRES;
        }
        x = Y;
    }
}

template <class PT>
CA_DEV void deinterleave_hadamard_via(x16_t *X, PT tmp, int N0, int stride, int hadamard)   // bands.c:524-549
{
    const int N = N0 * stride;
    const u8 *ordery = CLT_ordery_table + stride - 2;
    if (LANES == 1) {
        // one lane owns the frame: walk the N outputs flat, eight loads in flight before the first store
        int sidx = 0, j = 0, d = hadamard ? ordery[0] : 0;
        for (int k = 0; k < N; k += 8) {
            i32 v[8];
            int dst[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const bool ok = k + u < N;
                v[u] = ok ? (i32)X[j * stride + sidx] : 0;
                dst[u] = ok ? d * N0 + j : -1;
                if (++j == N0) { j = 0; sidx++; d = hadamard ? ordery[sidx < stride ? sidx : 0] : sidx; }
            }
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (dst[u] >= 0) tmp[dst[u]] = (i16)v[u];
        }
    } else {
        for (int s = 0; s < stride; s++) {
            int d = hadamard ? ordery[s] : s;
            for (int j = lane(); j < N0; j += LANES) tmp[d * N0 + j] = X[j * stride + s];
        }
    }
    wave_sync();
    CA_UNROLL_LANE
    for (int k = lane(); k < N; k += LANES) X[k] = tmp[k];
    wave_sync();
}

template <class L>
CA_DEV void deinterleave_hadamard_wave(L &F, x16_t *X, int N0, int stride, int hadamard)
{
#if defined(CA_LANE_FRAME)
    // the PVQ search scratch is idle here: bands of up to 2*PVQ_LDS_N bins bounce through its LDS copy
    if (N0 * stride <= 2 * PVQ_LDS_N) deinterleave_hadamard_via(X, lds_col(F.lds_pvq16), N0, stride, hadamard);
    else deinterleave_hadamard_via(X, priv((i16 *)F.s.pvq.xabs), N0, stride, hadamard);
#else
    deinterleave_hadamard_via(X, (i16 *)F.s.pvq.xabs, N0, stride, hadamard);
#endif
}

// quant_band (bands.c:1044-1174) + quant_partition (bands.c:864-1042), encode only, no lowband.
// The reference recurses (a partition splits into two half-size partitions, LM 3 -> -1, depth <= 4); here
// the recursion is an explicit depth-first walk: when a node splits, its second child is parked in a
// 4-entry LDS stack (F.pstack) and revived -- with the re-balanced bit budget, bands.c:961-981 -- once the
// first child's subtree has been coded. One call site of alg_quant, no device-side recursion.
template <class L>
CA_DEV void quant_band_wave(L &F, RangeEnc &ec, BandCtx &ctx, x16_t *Xband, int N, int b, int B, int LM)
{
    CA_STAMP_F(F, 22);
    CA_COUNT("quant_band", N);
    int N_B = (int)((u32)N / (u32)B);
    const int longBlocks = B == 1;
    int tf_change = ctx.tf_change;
    if (N == 1) { quant_band_n1_wave(ec, ctx, Xband, nullptr); return; }
    int recombine = tf_change > 0 ? tf_change : 0;
    CA_COUNT("band.tf_change", tf_change);
    CA_COUNT(B > 1 ? "band.short" : "band.long", N);
    CA_STAMP_F(F, 26);
#if defined(CA_LANE_FRAME)
    // Lane build: the time-frequency re-arrangement of a band (haar1 levels, then the de-interleave) is pure data movement
    // over N 16-bit bins. Done in place on X in HBM it costs every lane 2-byte loads and stores at a 4.7 KB stride from
    // its neighbours' (each touching its own cache line) -- ~190 of them per band. The band is instead pulled into the
    // idle per-lane scratch in LDS (all bands but the last fit: N <= LANE_SCRATCH_N = 144) with 16-byte loads, transformed
    // there, and written back once, in output order, with 16-byte stores.
    if (N <= LANE_SCRATCH_N && (N & 7) == 0 && (recombine > 0 || B > 1 || ((N_B & 1) == 0 && tf_change < 0))) {
        LdsCol<i16> T = lds_col(F.lds_pvq16);
        for (int k = 0; k < N; k += 8) {
            const v4i v = *reinterpret_cast<const CA_AS_GLB v4i *>(Xband + k);
            T[k + 0] = (i16)v.x; T[k + 1] = (i16)(v.x >> 16); T[k + 2] = (i16)v.y; T[k + 3] = (i16)(v.y >> 16);
            T[k + 4] = (i16)v.z; T[k + 5] = (i16)(v.z >> 16); T[k + 6] = (i16)v.w; T[k + 7] = (i16)(v.w >> 16);
        }
        wave_sync();
        for (int k = 0; k < recombine; k++) haar1_wave(T, N >> k, 1 << k);
        CA_STAMP_F(F, 27);
        B >>= recombine;
        N_B <<= recombine;
        while ((N_B & 1) == 0 && tf_change < 0) {
            haar1_wave(T, N_B, B);
            B <<= 1;
            N_B >>= 1;
            tf_change++;
        }
        CA_STAMP_F(F, 28);
        // write-back in output order; with B > 1 through the de-interleave (bands.c:524-549): output d*N0 + j takes
        // input j*stride + s, d = ordery[s] for the Hadamard ordering of long blocks, d = s otherwise
        const int stride = B > 1 ? B << recombine : 1, N0 = B > 1 ? N_B >> recombine : N;
        const u8 *ordery = CLT_ordery_table + stride - 2;
        int d = 0, j = 0, sidx = 0;
        if (stride > 1 && longBlocks) { while (ordery[sidx] != 0) sidx++; }
        for (int k = 0; k < N; k += 8) {
            u32 w[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                w[u] = (u32)(u16)T[j * stride + sidx];
                if (++j == N0) {
                    j = 0;
                    d++;
                    sidx = d;
                    if (stride > 1 && longBlocks && d < stride) { sidx = 0; while (ordery[sidx] != d) sidx++; }
                }
            }
            v4i v;
            v.x = (i32)(w[0] | (w[1] << 16)); v.y = (i32)(w[2] | (w[3] << 16));
            v.z = (i32)(w[4] | (w[5] << 16)); v.w = (i32)(w[6] | (w[7] << 16));
            *reinterpret_cast<CA_AS_GLB v4i *>(Xband + k) = v;
        }
        wave_sync();
        recombine = 0;
        tf_change = 0;                 // nothing left for the in-place path below
        if (B > 1) B = -B;             // (restored just below) marks the de-interleave as done
    }
#endif
    for (int k = 0; k < recombine; k++) { CA_COUNT("band.haar_recombine", N); haar1_wave(Xband, N >> k, 1 << k); }
    CA_STAMP_F(F, 27);
    B >>= recombine;
    N_B <<= recombine;
    while ((N_B & 1) == 0 && tf_change < 0) {
        CA_COUNT("band.haar_timediv", N);
        haar1_wave(Xband, N_B, B);
        B <<= 1;
        N_B >>= 1;
        tf_change++;
    }
    CA_STAMP_F(F, 28);
    bool deinterleaved = false;
    if (B < 0) { B = -B; deinterleaved = true; }
    const int B0band = B;
    if (B0band > 1) CA_COUNT("band.deinterleave", N);
    if (B0band > 1 && !deinterleaved) deinterleave_hadamard_wave(F, Xband, N_B >> recombine, B0band << recombine, longBlocks);
    CA_STAMP_F(F, 21);

    int sp = 0;
    int xoff = 0;
    for (;;) {
        // descend: split until the current node is a leaf. Written as an inner loop so that, in the
        // lane-per-frame build, the lanes of a wavefront split together and then code their leaves together.
        while (LM != -1 && N > 2 && b > pulse_cache_max(ctx.i, LM) + 12) {
            x16_t *X = Xband + xoff;
            const int B0 = B;
            N >>= 1;
            x16_t *Y = X + N;
            LM -= 1;
            B = (B + 1) >> 1;
            SplitCtx sc = compute_theta_wave(F, ec, ctx, X, Y, N, &b, B, B0, LM, 0);
            int delta = sc.delta;
            const int itheta = sc.itheta;
            if (B0 > 1 && (itheta & 0x3fff)) {
                if (itheta > 8192) delta -= delta >> (4 - LM);
                else delta = imin(0, delta + (N << BITRES >> (5 - LM)));
            }
            const int mbits = imax(0, imin(b, (b - delta) / 2));
            const int sbits = b - mbits;
            ctx.remaining_bits -= sc.qalloc;
            const int mid_first = mbits >= sbits;
            // park the second child: {xoff, bits, N, B, LM, remaining_bits at the split, first child's bits, re-balance allowed}
            if (lane() == 0) {
                i32 *fr = F.pstack[sp];
                fr[0] = mid_first ? xoff + N : xoff;
                fr[1] = mid_first ? sbits : mbits;
                fr[2] = N;
                fr[3] = B;
                fr[4] = LM;
                fr[5] = ctx.remaining_bits;
                fr[6] = mid_first ? mbits : sbits;
                fr[7] = mid_first ? (itheta != 0) : (itheta != 16384);
            }
            sp++;
            wave_sync();
            if (!mid_first) xoff += N;
            b = mid_first ? mbits : sbits;
        }
        x16_t *X = Xband + xoff;
        // leaf: the basic no-split case (bands.c:983-1039)
        CA_STAMP_F(F, 24);
        int q = bits2pulses(ctx.i, LM, b);
        CA_COUNT(q ? "node.leaf" : "node.leaf_q0", N);
        int curr_bits = pulses2bits(ctx.i, LM, q);
        ctx.remaining_bits -= curr_bits;
        while (ctx.remaining_bits < 0 && q > 0) {
            ctx.remaining_bits += curr_bits;
            q--;
            curr_bits = pulses2bits(ctx.i, LM, q);
            ctx.remaining_bits -= curr_bits;
        }
        if (q != 0) alg_quant_wave(F, ec, X, N, get_pulses(q), ctx.spread, B);
        if (sp == 0) break;
        sp--;
        const i32 *fr = F.pstack[sp];
        xoff = uni(fr[0]);
        b = uni(fr[1]);
        N = uni(fr[2]);
        B = uni(fr[3]);
        LM = uni(fr[4]);
        i32 rebalance = uni(fr[6]) - (uni(fr[5]) - ctx.remaining_bits);
        if (rebalance > 3 << BITRES && uni(fr[7])) b += rebalance - (3 << BITRES);
    }
}

#if defined(CA_LANE_FRAME)
// The non-stereo half of compute_theta (bands.c:645-817 with stereo == 0: the split of a partition into two halves),
// on the band in LDS. X / Y are read, not modified.
template <class L>
CA_DEVFN SplitCtx split_theta_lane(L &F, RangeEnc &ec, BandCtx &ctx, CA_AS_LDS const i16 *X, CA_AS_LDS const i16 *Y, int N, int *b,
                                   int B0, int LM)
{
    SplitCtx sc;
    CA_STAMP_F(F, 22);
    const int pulse_cap = CLT_logN400[ctx.i] + LM * (1 << BITRES);
    const int offset = (pulse_cap >> 1) - QTHETA_OFFSET;
    const int qn = compute_qn(N, *b, offset, pulse_cap, 0);
    // stereo_itheta(X, Y, 0, N) (vq.c:376-408)
    i32 pm = 0, ps = 0;
    for (int k = 0; k < N; k += 8, X += 8 * LDS_COL, Y += 8 * LDS_COL) {
        i32 xv[8], yv[8];
        const int rem = N - k;
#pragma unroll
        for (int u = 0; u < 8; u++) { xv[u] = X[u * LDS_COL]; yv[u] = Y[u * LDS_COL]; }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const i32 x = u < rem ? xv[u] : 0, y = u < rem ? yv[u] : 0;
            pm = mac16_16(pm, x, x);
            ps = mac16_16(ps, y, y);
        }
    }
    const i32 mid = (i16)celt_sqrt(add32(1, pm)), side = (i16)celt_sqrt(add32(1, ps));
    int itheta = mul16_16_q15(20861, celt_atan2p(side, mid));                     // QCONST16(0.63662f,15)
    const i32 tell = (i32)ec_tell_frac(ec);
    if (qn != 1) {
        itheta = (itheta * qn + 8192) >> 14;
        if (B0 > 1) {
            ec_enc_uint(ec, (u32)itheta, (u32)(qn + 1));
        } else {
            const int ft = ((qn >> 1) + 1) * ((qn >> 1) + 1);
            const int fs = itheta <= (qn >> 1) ? itheta + 1 : qn + 1 - itheta;
            const int fl = itheta <= (qn >> 1) ? (itheta * (itheta + 1)) >> 1 : ft - (((qn + 1 - itheta) * (qn + 2 - itheta)) >> 1);
            ec_encode(ec, (u32)fl, (u32)(fl + fs), (u32)ft);
        }
        itheta = (int)((u32)(itheta * 16384) / (u32)qn);
    }
    const int qalloc = (int)((i32)ec_tell_frac(ec) - tell);
    *b -= qalloc;
    int imid, iside, delta;
    if (itheta == 0) { imid = 32767; iside = 0; delta = -16384; }
    else if (itheta == 16384) { imid = 0; iside = 32767; delta = 16384; }
    else {
        imid = bitexact_cos((i16)itheta);
        iside = bitexact_cos((i16)(16384 - itheta));
        delta = frac_mul16((N - 1) << 7, bitexact_log2tan(iside, imid));
    }
    CA_STAMP_F(F, 16);
    sc.inv = 0; sc.imid = imid; sc.iside = iside; sc.delta = delta; sc.itheta = itheta; sc.qalloc = qalloc;
    return sc;
}

// quant_band + quant_partition of the lane build: the band is pulled into this lane's LDS column ONCE (16-byte loads) and
// everything that follows -- the haar1 levels, the de-interleave, the inner products of every split, the leaves -- works on
// that copy with ds_read / ds_write; nothing goes back to HBM (the encoder never reads X again after coding it, bands.c
// resynth == 0). Layout of the column: celt_enc_back.h LANE_HALF / LANE_IY. Bands of up to 96 bins de-interleave from one
// half into the other and then have the free half for the leaf copy; the two widest bands (144 / 176 bins) lie across
// both halves, de-interleave through their own rows in HBM and keep 64 slots for leaves of up to 32 bins. A leaf that
// does not fit (an unsplit wide band: few pulses over many bins) is searched in private memory by the generic body.
// N bins (a multiple of eight) of this lane's row of X into its column, eight 16-byte loads issued before the first of
// them is waited for: N is uniform (a band's width), so the guards are scalar branches. (A loop of load - eight column
// stores per chunk, unrolled or not, was compiled as one exposed memory round trip per chunk: the staging of the 42
// band-channels of a frame cost 240 of them, 6 % of the kernel.)
CA_DEV void stage_band_lane(const x16_t *Xband, CA_AS_LDS i16 *q, int N)
{
    for (int k = 0; k < N; k += 64, q += 64 * LDS_COL) {
        v4i v[8];
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (k + 8 * u < N) v[u] = *reinterpret_cast<const CA_AS_GLB v4i *>(Xband + k + 8 * u);
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (k + 8 * u < N) {
                CA_AS_LDS i16 *c = q + 8 * u * LDS_COL;
                c[0 * LDS_COL] = (i16)v[u].x; c[1 * LDS_COL] = (i16)(v[u].x >> 16); c[2 * LDS_COL] = (i16)v[u].y; c[3 * LDS_COL] = (i16)(v[u].y >> 16);
                c[4 * LDS_COL] = (i16)v[u].z; c[5 * LDS_COL] = (i16)(v[u].z >> 16); c[6 * LDS_COL] = (i16)v[u].w; c[7 * LDS_COL] = (i16)(v[u].w >> 16);
            }
    }
}

template <class L>
CA_DEV void quant_band_lane(L &F, RangeEnc &ec, BandCtx &ctx, x16_t *Xband, int N, int b, int B, int LM, int staged = 0)
{
    CA_STAMP_F(F, 22);
    CA_AS_LDS i16 *const S = F.col;                            // slot 0 of this lane's column
    const int Nband = N;
    int N_B = (int)((u32)N / (u32)B);
    const int longBlocks = B == 1;
    int tf_change = ctx.tf_change;
    if (N == 1) { quant_band_n1_wave(ec, ctx, Xband, nullptr); return; }
    int recombine = tf_change > 0 ? tf_change : 0;
    CA_STAMP_F(F, 26);
    if (!staged) stage_band_lane(Xband, S, N);         // (staged: compute_theta_wave's stereo split left this child in the column)
    for (int k = 0; k < recombine; k++) haar1_wave(lds_col(S), N >> k, 1 << k);
    CA_STAMP_F(F, 27);
    B >>= recombine;
    N_B <<= recombine;
    while ((N_B & 1) == 0 && tf_change < 0) {
        haar1_wave(lds_col(S), N_B, B);
        B <<= 1;
        N_B >>= 1;
        tf_change++;
    }
    CA_STAMP_F(F, 28);
    CA_AS_LDS i16 *cur = S;                                    // where the band lives from here on
    if (B > 1) {
        // deinterleave_hadamard (bands.c:524-549): output d*N0 + j takes input j*stride + s, d = ordery[s] for the Hadamard
        // ordering of long blocks, d = s otherwise; walked in output order
        const int stride = B << recombine, N0 = N_B >> recombine;
        const u8 *ordery = CLT_ordery_table + stride - 2;
        int d = 0, j = 0, sidx = 0;
        if (longBlocks) { while (ordery[sidx] != 0) sidx++; }
        const bool narrow = Nband <= LANE_HALF;
        CA_AS_LDS i16 *dst = S + LANE_HALF * LDS_COL;
        for (int k = 0; k < N; k += 8) {
            u32 w[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                w[u] = (u32)(u16)S[(j * stride + sidx) * LDS_COL];
                if (++j == N0) {
                    j = 0;
                    d++;
                    sidx = d;
                    if (longBlocks && d < stride) { sidx = 0; while (ordery[sidx] != d) sidx++; }
                }
            }
            if (narrow) {
#pragma unroll
                for (int u = 0; u < 8; u++) dst[u * LDS_COL] = (i16)w[u];
                dst += 8 * LDS_COL;
            } else {
                v4i v;
                v.x = (i32)(w[0] | (w[1] << 16)); v.y = (i32)(w[2] | (w[3] << 16));
                v.z = (i32)(w[4] | (w[5] << 16)); v.w = (i32)(w[6] | (w[7] << 16));
                *reinterpret_cast<CA_AS_GLB v4i *>(Xband + k) = v;
            }
        }
        if (narrow) {
            cur = S + LANE_HALF * LDS_COL;
        } else {
            stage_band_lane(Xband, S, N);
        }
    }
    CA_STAMP_F(F, 21);
    // scratch for the leaves: copy of the leaf + pulse counts
    const int leaf_max = Nband <= LANE_HALF ? LANE_IYN : 32;
    CA_AS_LDS i16 *const leaf = Nband <= LANE_HALF ? (cur == S ? S + LANE_HALF * LDS_COL : S) : S + 176 * LDS_COL;
    CA_AS_LDS u16 *const cnt = (CA_AS_LDS u16 *)(Nband <= LANE_HALF ? S + LANE_IY * LDS_COL : S + 208 * LDS_COL);

    int sp = 0;
    int xoff = 0;
    for (;;) {
        // descend: split until the current node is a leaf
        while (LM != -1 && N > 2 && b > pulse_cache_max(ctx.i, LM) + 12) {
            const int B0 = B;
            N >>= 1;
            LM -= 1;
            B = (B + 1) >> 1;
            SplitCtx sc = split_theta_lane(F, ec, ctx, cur + xoff * LDS_COL, cur + (xoff + N) * LDS_COL, N, &b, B0, LM);
            int delta = sc.delta;
            const int itheta = sc.itheta;
            if (B0 > 1 && (itheta & 0x3fff)) {
                if (itheta > 8192) delta -= delta >> (4 - LM);
                else delta = imin(0, delta + (N << BITRES >> (5 - LM)));
            }
            const int mbits = imax(0, imin(b, (b - delta) / 2));
            const int sbits = b - mbits;
            ctx.remaining_bits -= sc.qalloc;
            const int mid_first = mbits >= sbits;
            // park the second child: {xoff, N, B, LM, re-balance allowed} packed, its bits, and (first child's bits - remaining_bits
            // at the split): rebalance (bands.c:961-981) = that + remaining_bits when the child is revived
            {
                const u32 w0 = (u32)(mid_first ? xoff + N : xoff) | ((u32)N << 8) | ((u32)B << 16) | ((u32)(LM + 1) << 21)
                             | ((u32)(mid_first ? (itheta != 0) : (itheta != 16384)) << 23);
                const u32 w1 = (u32)(mid_first ? sbits : mbits), w2 = (u32)((mid_first ? mbits : sbits) - ctx.remaining_bits);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    F.ps0[k] = sp == k ? w0 : F.ps0[k];
                    F.ps1[k] = sp == k ? w1 : F.ps1[k];
                    F.ps2[k] = sp == k ? w2 : F.ps2[k];
                }
            }
            sp++;
            if (!mid_first) xoff += N;
            b = mid_first ? mbits : sbits;
        }
        // leaf: the basic no-split case (bands.c:983-1039)
        int q = bits2pulses(ctx.i, LM, b);
        int curr_bits = pulses2bits(ctx.i, LM, q);
        ctx.remaining_bits -= curr_bits;
        while (ctx.remaining_bits < 0 && q > 0) {
            ctx.remaining_bits += curr_bits;
            q--;
            curr_bits = pulses2bits(ctx.i, LM, q);
            ctx.remaining_bits -= curr_bits;
        }
        if (q != 0) {
            const int K = get_pulses(q);
            CA_STAMP_F(F, 25);
            CA_AS_LDS const i16 *src = cur + xoff * LDS_COL;
            if (N <= leaf_max) {
                CA_AS_LDS i16 *dst = leaf;
                for (int k = 0; k < N; k += 8, src += 8 * LDS_COL, dst += 8 * LDS_COL) {
                    i32 v[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) v[u] = src[u * LDS_COL];       // reads up to 7 slots past the leaf: inside the column
#pragma unroll
                    for (int u = 0; u < 8; u++) dst[u * LDS_COL] = (i16)v[u];
                }
                alg_quant_lane(F, ec, leaf, cnt, N, K, ctx.spread, B);
            } else if (Nband <= LANE_HALF) {
                // an unsplit band of 64 or 96 bins: searched where it lies (its bins are dead once they are coded: resynth == 0),
                // the pulse counts in the band's other buffer. (Through the generic body below, whose search state is in HBM, a
                // handful of such leaves per wavefront cost every lane of it ~25 k cycles each.)
                CA_COUNT("lane.wide_leaf_inplace", N);
                alg_quant_lane(F, ec, const_cast<CA_AS_LDS i16 *>(src), (CA_AS_LDS u16 *)(cur == S ? S + LANE_HALF * LDS_COL : S), N, K,
                               ctx.spread, B);
            } else if (N <= 48) {
                // 36 / 44 bins, a quarter of one of the two widest bands: in place as well, the pulse counts in slots 184..231
                // (the narrow leaves' copy and count slots, 176..239, are idle); the quantiser zeroes the slots up to the next
                // multiple of eight past the leaf, which are the first bins of the next partition: kept aside and put back
                CA_COUNT("lane.wide_leaf_inplace", N);
                CA_AS_LDS i16 *const x = const_cast<CA_AS_LDS i16 *>(src);
                CA_AS_LDS i16 *const tail = x + (N & ~7) * LDS_COL;
                i32 keep[8];
#pragma unroll
                for (int u = 0; u < 8; u++) keep[u] = tail[u * LDS_COL];
                alg_quant_lane(F, ec, x, (CA_AS_LDS u16 *)(S + 184 * LDS_COL), N, K, ctx.spread, B);
                const int r0 = N & 7;
#pragma unroll
                for (int u = 0; u < 8; u++)
                    if (u >= r0) tail[u * LDS_COL] = (i16)keep[u];
            } else {
                // a leaf wider than the column's leaf slots (72 .. 176 bins of the two widest bands: a handful of pulses; N is a
                // multiple of eight): searched in place in the band buffer by the generic body, its search state (2*iy, |x|, iy:
                // 3 N 16-bit values) in bins of X this frame has already coded -- bands below this one are dead in both channels
                // (resynth == 0) and the first band with such a leaf starts at bin 320 of a channel
                x16_t *const gs = F.x16;
                CA_COUNT("lane.wide_leaf", N);
                alg_quant_body(F, ec, lds_col(const_cast<CA_AS_LDS i16 *>(src)), gs, gs + N, gs + 2 * N, N, K, ctx.spread, B);
            }
        }
        if (sp == 0) break;
        sp--;
        const u32 w0 = sp == 0 ? F.ps0[0] : sp == 1 ? F.ps0[1] : sp == 2 ? F.ps0[2] : F.ps0[3];
        const u32 w1 = sp == 0 ? F.ps1[0] : sp == 1 ? F.ps1[1] : sp == 2 ? F.ps1[2] : F.ps1[3];
        const u32 w2 = sp == 0 ? F.ps2[0] : sp == 1 ? F.ps2[1] : sp == 2 ? F.ps2[2] : F.ps2[3];
        xoff = (int)(w0 & 255u);
        N = (int)((w0 >> 8) & 255u);
        B = (int)((w0 >> 16) & 31u);
        LM = (int)((w0 >> 21) & 3u) - 1;
        b = (i32)w1;
        const i32 rebalance = (i32)w2 + ctx.remaining_bits;
        if (rebalance > 3 << BITRES && ((w0 >> 23) & 1u)) b += rebalance - (3 << BITRES);
    }
}
#endif

// quant_all_bands(encode = 1, start 0, end 21, LM 3)  (bands.c:1337-1502) with quant_band_stereo
// (bands.c:1176-1335) folded in: per band up to two quant_band jobs (mid/side or L/R) run through ONE
// call site, the second with the re-balanced budget.
template <class L>
CA_DEV void quant_all_bands_wave(L &F, RangeEnc &ec, int C, int shortBlocks, int spread, int dual_stereo,
                                 int intensity, i32 total_bits, i32 balance, int codedBands)
{
    const int LM = LM3, M = M8;
    const int B = shortBlocks ? M : 1;
    x16_t *X_ = frame_X(F), *Y_ = C == 2 ? X_ + FRAME : nullptr;
    BandCtx ctx;
    ctx.intensity = intensity;
    ctx.spread = spread;
    for (int i = 0; i < NB; i++) {
        CA_STAMP_F(F, 29);
        ctx.i = i;
        x16_t *X = X_ + M * CLT_eband5ms[i];
        x16_t *Y = Y_ ? Y_ + M * CLT_eband5ms[i] : nullptr;
        const int N = M * CLT_eband5ms[i + 1] - M * CLT_eband5ms[i];
        i32 tell = (i32)ec_tell_frac(ec);
        if (i != 0) balance -= tell;
        i32 remaining_bits = total_bits - tell - 1;
        ctx.remaining_bits = remaining_bits;
        int b;
        if (i <= codedBands - 1) {
            i32 curr_balance = balance / imin(3, codedBands - i);
            b = imax(0, imin(16383, imin(remaining_bits + 1, frame_pulses(F, i) + curr_balance)));
        } else {
            b = 0;
        }
        ctx.tf_change = frame_tf_change(F, i);
        if (dual_stereo && i == intensity) dual_stereo = 0;

        CA_STAMP_F(F, 23);
        // plan the jobs
        int njobs = 0, rebal = 0, allow2 = 0, staged0 = 0;
        (void)staged0;
        x16_t *jx0 = X, *jx1 = Y;
        int jb0 = b, jb1 = 0;
        if (dual_stereo) {
            njobs = 2; jb0 = b / 2; jb1 = b / 2;
        } else if (Y) {
            if (N == 1) {
                quant_band_n1_wave(ec, ctx, X, Y);
            } else {
#if defined(CA_LANE_FRAME)
                SplitCtx sc = compute_theta_wave(F, ec, ctx, X, Y, N, &b, B, B, LM, 1, 1);
                staged0 = sc.staged;
#else
                SplitCtx sc = compute_theta_wave(F, ec, ctx, X, Y, N, &b, B, B, LM, 1);
#endif
                const int itheta = sc.itheta;
                if (N == 2) {
                    int mbits = b, sbits = 0;
                    if (itheta != 0 && itheta != 16384) sbits = 1 << BITRES;
                    mbits -= sbits;
                    const int c = itheta > 8192;
                    ctx.remaining_bits -= sc.qalloc + sbits;
                    x16_t *x2 = c ? Y : X, *y2 = c ? X : Y;
                    if (sbits) {
                        int sign = (uni((i32)x2[0]) * uni((i32)y2[1]) - uni((i32)x2[1]) * uni((i32)y2[0])) < 0;
                        ec_enc_bits(ec, (u32)sign, 1);
                    }
                    njobs = 1; jx0 = x2; jb0 = mbits;
                } else {
                    int mbits = imax(0, imin(b, (b - sc.delta) / 2));
                    int sbits = b - mbits;
                    ctx.remaining_bits -= sc.qalloc;
                    njobs = 2; rebal = 1;
                    if (mbits >= sbits) { jx0 = X; jb0 = mbits; jx1 = Y; jb1 = sbits; allow2 = itheta != 0; }
                    else { jx0 = Y; jb0 = sbits; jx1 = X; jb1 = mbits; allow2 = itheta != 16384; }
                }
            }
        } else {
            njobs = 1;
        }
        const i32 rebalance0 = ctx.remaining_bits;
        for (int j = 0; j < njobs; j++) {
            x16_t *jx = j == 0 ? jx0 : jx1;
            int jb = j == 0 ? jb0 : jb1;
            if (j == 1 && rebal) {
                i32 rebalance = jb0 - (rebalance0 - ctx.remaining_bits);
                if (rebalance > 3 << BITRES && allow2) jb += rebalance - (3 << BITRES);
            }
#if defined(CA_LANE_FRAME)
            quant_band_lane(F, ec, ctx, jx, N, jb, B, LM, j == 0 && staged0);
#else
            quant_band_wave(F, ec, ctx, jx, N, jb, B, LM);
#endif
        }
        balance += frame_pulses(F, i) + tell;
    }
}

}  // namespace ca

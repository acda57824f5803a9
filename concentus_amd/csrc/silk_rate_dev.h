// silk_rate_dev.h -- the control arithmetic of silk_encode_frame_FIX's bitrate loop (opus-fix/silk/fixed/encode_frame_FIX.c:276-423,
// SURVEY 8f row 4, tenth slice): what happens between two quantise + entropy-code passes of one frame.
//
//   silk_encode_frame_FIX (the loop)          opus-fix/silk/fixed/encode_frame_FIX.c:263-423
//   silk_gains_ID                             opus-fix/silk/gain_quant.c:147-161
//   silk_gains_quant                          opus-fix/silk/gain_quant.c:41-103 (silk_gains_dev.h)
//
// The reference's loop body is  [top: reuse a bracket's bit count or restore + NSQ + encode]  [bottom: bracket / gain update].
// A step here starts where a coding pass ended: it runs the bottom half, then the top half of the next iteration, and keeps
// going while the next iteration needs no coding pass (its gains are those of a bracket already measured).
#pragma once
#include "silk_gains_dev.h"
#include "../../include/opusgpu_silk.h"

namespace ca {

CA_DEV i32 silk_gains_ID_dev(const i8 *ind, int nb_subfr)                                    // gain_quant.c:147-161
{
    i32 id = 0;
    for (int k = 0; k < nb_subfr; k++) id = s_addw((i32)ind[k], shl32(id, 8));
    return id;
}

// nBits: ec_tell() of the coder after the pass just coded.
// c: the frame's record (the loop's locals between passes); finished frames are left as they are.
CA_DEV void silk_rate_control_step_dev(opusgpu_silk_rate_ctl &c, i32 nBits)
{
    enum { MAX_ITER = 6 };
    c.recode = c.save2 = c.restore2 = 0;
    if (c.done) return;
    c.nBits = nBits; c.status = 0;
    if (!c.started) {                                                                       // :264-270
        c.started = 1; c.iter = 0; c.gainMult_Q8 = 256; c.found_lower = c.found_upper = 0;
        c.gainsID = silk_gains_ID_dev((const i8 *)c.GainsIndices, c.nb_subfr);
        c.gainsID_lower = c.gainsID_upper = -1;
        c.nBits_lower = c.nBits_upper = c.gainMult_lower = c.gainMult_upper = 0; c.LastGainIndex_copy2 = 0;
        c.passes = 0;
    }
    c.passes++;
    if (c.useCBR == 0 && c.iter == 0 && nBits <= c.maxBits) { c.done = 1; return; }          // :352-354
    for (;;) {
        if (c.iter == MAX_ITER) {                                                           // :358-369
            if (c.found_lower && (c.gainsID == c.gainsID_lower || nBits > c.maxBits)) { c.restore2 = 1; c.LastGainIndex = c.LastGainIndex_copy2; }
            c.done = 1;
            return;
        }
        if (nBits > c.maxBits) {                                                            // :371-382
            if (c.found_lower == 0 && c.iter >= 2) {
                c.Lambda_Q10 = s_addw(c.Lambda_Q10, c.Lambda_Q10 >> 1);
                c.found_upper = 0;
                c.gainsID_upper = -1;
            } else {
                c.found_upper = 1; c.nBits_upper = nBits; c.gainMult_upper = c.gainMult_Q8; c.gainsID_upper = c.gainsID;
            }
        } else if (nBits < c.maxBits - 5) {                                                 // :383-396
            c.found_lower = 1; c.nBits_lower = nBits; c.gainMult_lower = c.gainMult_Q8;
            if (c.gainsID != c.gainsID_lower) {
                c.gainsID_lower = c.gainsID;
                c.save2 = 1;                                                                // coder + quantiser state of the pass just coded
                c.LastGainIndex_copy2 = c.LastGainIndex;
            }
        } else {                                                                            // within 5 bits: close enough
            c.done = 1;
            return;
        }
        if ((c.found_lower & c.found_upper) == 0) {                                         // :401-409: the high-rate rate/distortion slope
            i32 gain_factor_Q16 = s_log2lin(shl32(nBits - c.maxBits, 7) / c.frame_length + 2048);
            gain_factor_Q16 = imin(gain_factor_Q16, 131072);
            if (nBits > c.maxBits) gain_factor_Q16 = imax(gain_factor_Q16, 85197);          // SILK_FIX_CONST(1.3, 16)
            c.gainMult_Q8 = (i16)s_smulwb(gain_factor_Q16, c.gainMult_Q8);
        } else {                                                                            // :410-420: interpolate between the brackets
            c.gainMult_Q8 = (i16)(c.gainMult_lower + (c.gainMult_upper - c.gainMult_lower) * (c.maxBits - c.nBits_lower) / (c.nBits_upper - c.nBits_lower));   // silk_DIV32_16 does not narrow
            const i32 hi = s_addw(c.gainMult_lower, (c.gainMult_upper - c.gainMult_lower) >> 2);
            const i32 lo = s_subw(c.gainMult_upper, (c.gainMult_upper - c.gainMult_lower) >> 2);
            if (c.gainMult_Q8 > hi) c.gainMult_Q8 = (i16)hi;
            else if (c.gainMult_Q8 < lo) c.gainMult_Q8 = (i16)lo;
        }
        for (int i = 0; i < c.nb_subfr; i++) c.Gains_Q16[i] = s_lshift_sat32(s_smulwb(c.GainsUnq_Q16[i], c.gainMult_Q8), 8);
        int lgi = c.lastGainIndexPrev;                                                      // :426-429
        silk_gains_quant_dev((i8 *)c.GainsIndices, (i32 *)c.Gains_Q16, &lgi, c.condCoding == 2, c.nb_subfr);
        c.LastGainIndex = lgi;
        c.gainsID = silk_gains_ID_dev((const i8 *)c.GainsIndices, c.nb_subfr);
        c.iter++;
        // top of the next iteration (:277-281)
        if (c.gainsID == c.gainsID_lower) nBits = c.nBits_lower;
        else if (c.gainsID == c.gainsID_upper) nBits = c.nBits_upper;
        else { c.recode = 1; return; }
    }
}

}  // namespace ca

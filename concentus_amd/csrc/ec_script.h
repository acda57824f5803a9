// ec_script.h -- a sequence of range-coder calls run as ONE device call.
//
// The reference has no hook slot for ec_enc_* / ec_dec_* (plain externs, celt/entenc.h, celt/entdec.h): every symbol is a few
// dozen instructions on the caller's ec_ctx, so a per-symbol launch would be all latency. The boundary offered instead is a
// script: the caller lists the entenc.h / entdec.h calls it would have made (opcode + up to three arguments each), the device runs
// them back to back on the ec_ctx it was handed and returns the coder (and, decoding, one value per call). That is also how the
// reference's own coder test (celt/tests/test_unit_entropy.c) is replayed against the device code (tests/test_ec_script_*.py).
//
//   encoder ops (entenc.c)                                     decoder ops (entdec.c)                        out[i]
//   0 ec_encode(fl, fh, ft)                         :187       16 ec_decode(ft)                      :156   the cumulative frequency
//   1 ec_encode_bin(fl, fh, bits)                   :218       17 ec_decode_bin(bits)                :176   "
//   2 ec_enc_bit_logp(val, logp)                    :249       18 ec_dec_update(fl, fh, ft)          :183   0
//   3 ec_enc_uint(fl, ft)                           :313       19 ec_dec_bit_logp(logp)              :207   the bit
//   4 ec_enc_bits(fl, bits)                         :346       20 ec_dec_uint(ft)                    :241   the value
//   5 ec_enc_patch_initial_bits(val, nbits)         :386       21 ec_dec_bits(bits)                  :282   the value
//   6 ec_enc_shrink(size)                           :427       22 ec_laplace_decode(fs, decay)  laplace.c:93   the value
//   7 ec_enc_done()                                 :447       23 ec_tell()                                  whole bits used
//   8 ec_laplace_encode(value, fs, decay)   laplace.c:38       24 ec_tell_frac()                             1/8 bits used
//   9 ec_enc_icdf(s, table, ftb): table 0 = trim_icdf, 1 = spread_icdf, 2 = tapset_icdf, 3 = small_energy_icdf  :279
//                                                              25 ec_dec_icdf(table, ftb)            :223   the symbol
#pragma once
#include "rangecoder.h"
#include "rangedec.h"

namespace ca {

enum { ECS_ENCODE = 0, ECS_ENCODE_BIN, ECS_BIT_LOGP, ECS_UINT, ECS_BITS, ECS_PATCH_INITIAL_BITS, ECS_SHRINK, ECS_DONE, ECS_LAPLACE,
       ECS_ICDF, ECS_ENC_OPS,
       ECS_DEC_DECODE = 16, ECS_DEC_DECODE_BIN, ECS_DEC_UPDATE, ECS_DEC_BIT_LOGP, ECS_DEC_UINT, ECS_DEC_BITS, ECS_DEC_LAPLACE,
       ECS_DEC_TELL, ECS_DEC_TELL_FRAC, ECS_DEC_ICDF, ECS_DEC_OPS };

CA_DEV const u8 *ec_script_icdf(int t)
{
    return t == 0 ? CLT_trim_icdf : t == 1 ? CLT_spread_icdf : t == 2 ? CLT_tapset_icdf : CLT_small_energy_icdf;
}

// true when every opcode and argument of the script is one the reference's functions accept (their celt_assert()s)
CA_DEV bool ec_enc_script_ok(const i32 *ops, int n)
{
    bool ok = true;
    for (int k = 0; k < n; k++) {
        const i32 op = ops[4 * k], a = ops[4 * k + 1], b = ops[4 * k + 2], c = ops[4 * k + 3];
        if (op == ECS_ENCODE) ok &= (u32)a < (u32)b && (u32)b <= (u32)c && (u32)c >= 1;
        else if (op == ECS_ENCODE_BIN) ok &= c >= 1 && c <= 16 && (u32)a < (u32)b && (u32)b <= (1u << c);
        else if (op == ECS_BIT_LOGP) ok &= (a == 0 || a == 1) && b >= 1 && b <= 16;
        else if (op == ECS_UINT) ok &= (u32)b > 1 && (u32)a < (u32)b;
        else if (op == ECS_BITS) ok &= b >= 1 && b <= 25 && ((u32)a >> b) == 0;
        else if (op == ECS_PATCH_INITIAL_BITS) ok &= b >= 0 && b <= 8 && ((u32)a >> b) == 0;
        else if (op == ECS_SHRINK) ok &= a >= 0;
        else if (op == ECS_DONE) ok &= true;
        else if (op == ECS_LAPLACE) ok &= b > 0 && b < 32768 && c >= 0 && c <= 16384;
        else if (op == ECS_ICDF) ok &= b >= 0 && b <= 3 && c >= 1 && c <= 8 && a >= 0 && a < (b == 0 ? 11 : b == 1 ? 4 : 3);
        else ok = false;
    }
    return ok;
}

CA_DEV void ec_enc_run_script(RangeEnc &e, const i32 *ops, int n)
{
    for (int k = 0; k < n; k++) {
        const i32 op = uni(ops[4 * k]), a = uni(ops[4 * k + 1]), b = uni(ops[4 * k + 2]), c = uni(ops[4 * k + 3]);
        switch (op) {
        case ECS_ENCODE: ec_encode(e, (u32)a, (u32)b, (u32)c); break;
        case ECS_ENCODE_BIN: ec_encode_bin(e, (u32)a, (u32)b, (u32)c); break;
        case ECS_BIT_LOGP: ec_enc_bit_logp(e, a, (u32)b); break;
        case ECS_UINT: ec_enc_uint(e, (u32)a, (u32)b); break;
        case ECS_BITS: ec_enc_bits(e, (u32)a, (u32)b); break;
        case ECS_PATCH_INITIAL_BITS: ec_enc_patch_initial_bits(e, (u32)a, (u32)b); break;
        case ECS_SHRINK: if ((u32)a + e.end_offs <= e.storage && e.offs + e.end_offs <= (u32)a) ec_enc_shrink(e, (u32)a); else e.error = -1; break;
        case ECS_DONE: ec_enc_done(e); break;
        case ECS_LAPLACE: { int v = a; ec_laplace_encode(e, v, (u32)b, c); break; }
        case ECS_ICDF: ec_enc_icdf(e, a, ec_script_icdf(b), (u32)c); break;
        default: e.error = -1; break;
        }
        wave_sync();
    }
}

CA_DEV bool ec_dec_script_ok(const i32 *ops, int n)
{
    bool ok = true;
    for (int k = 0; k < n; k++) {
        const i32 op = ops[4 * k], a = ops[4 * k + 1], b = ops[4 * k + 2], c = ops[4 * k + 3];
        if (op == ECS_DEC_DECODE) ok &= (u32)a >= 1;
        else if (op == ECS_DEC_DECODE_BIN) ok &= a >= 1 && a <= 16;
        else if (op == ECS_DEC_UPDATE) ok &= (u32)a < (u32)b && (u32)b <= (u32)c;
        else if (op == ECS_DEC_BIT_LOGP) ok &= a >= 1 && a <= 16;
        else if (op == ECS_DEC_UINT) ok &= (u32)a > 1;
        else if (op == ECS_DEC_BITS) ok &= a >= 1 && a <= 25;
        else if (op == ECS_DEC_LAPLACE) ok &= a > 0 && a < 32768 && b >= 0 && b <= 16384;
        else if (op == ECS_DEC_TELL || op == ECS_DEC_TELL_FRAC) ok &= true;
        else if (op == ECS_DEC_ICDF) ok &= a >= 0 && a <= 3 && b >= 1 && b <= 8;
        else ok = false;
        (void)c;
    }
    return ok;
}

template <class OUT>
CA_DEV void ec_dec_run_script(RangeDec &d, const i32 *ops, int n, OUT out)
{
    for (int k = 0; k < n; k++) {
        const i32 op = ops[4 * k], a = ops[4 * k + 1], b = ops[4 * k + 2], c = ops[4 * k + 3];
        i32 r = 0;
        switch (op) {
        case ECS_DEC_DECODE: r = (i32)ec_decode(d, (u32)a); break;
        case ECS_DEC_DECODE_BIN: r = (i32)ec_decode_bin(d, (u32)a); break;
        case ECS_DEC_UPDATE: ec_dec_update(d, (u32)a, (u32)b, (u32)c); break;
        case ECS_DEC_BIT_LOGP: r = ec_dec_bit_logp(d, (u32)a); break;
        case ECS_DEC_UINT: r = (i32)ec_dec_uint(d, (u32)a); break;
        case ECS_DEC_BITS: r = (i32)ec_dec_bits(d, (u32)a); break;
        case ECS_DEC_LAPLACE: r = ec_laplace_decode(d, (u32)a, b); break;
        case ECS_DEC_TELL: r = ec_tell(d); break;
        case ECS_DEC_TELL_FRAC: r = (i32)ec_tell_frac(d); break;
        case ECS_DEC_ICDF: r = ec_dec_icdf(d, ec_script_icdf(a), (u32)b); break;
        default: d.error = -1; break;
        }
        out[k] = r;
    }
}

}  // namespace ca

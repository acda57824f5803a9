// rangedec.h -- the range decoder (opus-fix/celt/entdec.c:93-317, celt/laplace.c:93-134), the mirror of
// rangecoder.h. Written for one lane per frame: the state is ordinary per-lane data.
#pragma once
#include "rangecoder.h"

namespace ca {

struct RangeDec {
    const u8 *buf;
    u32 storage;
    u32 end_offs;
    u32 end_window;
    int nend_bits;
    int nbits_total;
    u32 offs;
    u32 rng;
    u32 val;
    u32 ext;
    int rem;
    int error;
};

CA_DEV int ec_read_byte(RangeDec &d) { return d.offs < d.storage ? d.buf[d.offs++] : 0; }                    // entdec.c:93
CA_DEV int ec_read_byte_from_end(RangeDec &d) { return d.end_offs < d.storage ? d.buf[d.storage - ++d.end_offs] : 0; }

CA_DEV void ec_dec_normalize(RangeDec &d)                                                                    // entdec.c:104-131
{
    while (d.rng <= CA_EC_CODE_BOT) {
        d.nbits_total += EC_SYM_BITS;
        d.rng <<= EC_SYM_BITS;
        int sym = d.rem;
        d.rem = ec_read_byte(d);
        sym = (sym << EC_SYM_BITS | d.rem) >> (EC_SYM_BITS - 7);                 // EC_CODE_EXTRA = 7
        d.val = ((d.val << EC_SYM_BITS) + (u32)(EC_SYM_MAX & ~sym)) & (CA_EC_CODE_TOP - 1);
    }
}

CA_DEV void ec_dec_init(RangeDec &d, const u8 *buf, u32 storage)                                             // entdec.c:133-154
{
    d.buf = buf;
    d.storage = storage;
    d.end_offs = 0;
    d.end_window = 0;
    d.nend_bits = 0;
    d.nbits_total = EC_CODE_BITS + 1 - ((EC_CODE_BITS - 7) / EC_SYM_BITS) * EC_SYM_BITS;
    d.offs = 0;
    d.rng = 1u << 7;
    d.rem = ec_read_byte(d);
    d.val = d.rng - 1 - (u32)(d.rem >> (EC_SYM_BITS - 7));
    d.ext = 0;
    d.error = 0;
    ec_dec_normalize(d);
}

CA_DEV int ec_tell(const RangeDec &d) { return d.nbits_total - ec_ilog(d.rng); }

CA_DEV u32 ec_tell_frac(const RangeDec &d)
{
    u32 nbits = (u32)d.nbits_total << 3;
    int l = ec_ilog(d.rng);
    u32 r = d.rng >> (l - 16);
    u32 b = (r >> 12) - 8;
    b += r > CLT_tell_frac_correction[b];
    return nbits - (u32)((l << 3) + (int)b);
}

CA_DEV u32 ec_decode(RangeDec &d, u32 ft)                                                                    // entdec.c:156
{
    d.ext = d.rng / ft;
    u32 s = d.val / d.ext;
    return ft - (s + 1 < ft ? s + 1 : ft);
}

CA_DEV u32 ec_decode_bin(RangeDec &d, u32 bits)                                                              // entdec.c:176
{
    d.ext = d.rng >> bits;
    u32 s = d.val / d.ext;
    return (1u << bits) - (s + 1u < (1u << bits) ? s + 1u : (1u << bits));
}

CA_DEV void ec_dec_update(RangeDec &d, u32 fl, u32 fh, u32 ft)                                               // entdec.c:183
{
    u32 s = d.ext * (ft - fh);
    d.val -= s;
    d.rng = fl > 0 ? d.ext * (fh - fl) : d.rng - s;
    ec_dec_normalize(d);
}

CA_DEV int ec_dec_bit_logp(RangeDec &d, u32 logp)                                                            // entdec.c:207
{
    u32 r = d.rng, v = d.val, s = r >> logp;
    int ret = v < s;
    if (!ret) d.val = v - s;
    d.rng = ret ? s : r - s;
    ec_dec_normalize(d);
    return ret;
}

CA_DEV int ec_dec_icdf(RangeDec &d, const u8 *icdf, u32 ftb)                                                 // entdec.c:223
{
    u32 s = d.rng, v = d.val, r = s >> ftb, t;
    int ret = -1;
    do {
        t = s;
        s = r * icdf[++ret];
    } while (v < s);
    d.val = v - s;
    d.rng = t - s;
    ec_dec_normalize(d);
    return ret;
}

CA_DEV u32 ec_dec_bits(RangeDec &d, u32 bits)                                                                // entdec.c:282
{
    u32 window = d.end_window;
    int available = d.nend_bits;
    if ((u32)available < bits) {
        do {
            window |= (u32)ec_read_byte_from_end(d) << available;
            available += EC_SYM_BITS;
        } while (available <= EC_WINDOW_SIZE - EC_SYM_BITS);
    }
    u32 ret = window & ((1u << bits) - 1u);
    window >>= bits;
    available -= (int)bits;
    d.end_window = window;
    d.nend_bits = available;
    d.nbits_total += (int)bits;
    return ret;
}

CA_DEV u32 ec_dec_uint(RangeDec &d, u32 ft)                                                                  // entdec.c:241
{
    ft--;
    int ftb = ec_ilog(ft);
    if (ftb > EC_UINT_BITS) {
        ftb -= EC_UINT_BITS;
        u32 ft1 = (ft >> ftb) + 1;
        u32 s = ec_decode(d, ft1);
        ec_dec_update(d, s, s + 1, ft1);
        u32 t = s << ftb | ec_dec_bits(d, (u32)ftb);
        if (t <= ft) return t;
        d.error = 1;
        return ft;
    }
    ft++;
    u32 s = ec_decode(d, ft);
    ec_dec_update(d, s, s + 1, ft);
    return s;
}

CA_DEV int ec_laplace_decode(RangeDec &d, u32 fs, int decay)                                                 // laplace.c:93-134
{
    int val = 0;
    u32 fm = ec_decode_bin(d, 15), fl = 0;
    if (fm >= fs) {
        val++;
        fl = fs;
        fs = (((32768 - 32 - fs) * (u32)(16384 - decay)) >> 15) + 1;             // ec_laplace_get_freq1 + LAPLACE_MINP
        while (fs > 1 && fm >= fl + 2 * fs) {
            fs *= 2;
            fl += fs;
            fs = ((fs - 2) * (u32)decay) >> 15;
            fs += 1;
            val++;
        }
        if (fs <= 1) {
            int di = (int)(fm - fl) >> 1;                                         // LAPLACE_LOG_MINP + 1
            val += di;
            fl += (u32)(2 * di);
        }
        if (fm < fl + fs) val = -val;
        else fl += fs;
    }
    u32 fh = fl + fs < 32768u ? fl + fs : 32768u;
    ec_dec_update(d, fl, fh, 32768);
    return val;
}

// One spelling for "code this decision": the encoder writes `val` and returns it, the decoder ignores it and
// returns what it reads (lets compute_allocation serve both directions, rate.c:391-447).
CA_DEV int coder_bit_logp(RangeEnc &e, int val, u32 logp) { ec_enc_bit_logp(e, val, logp); return val; }
CA_DEV int coder_bit_logp(RangeDec &d, int, u32 logp) { return ec_dec_bit_logp(d, logp); }
CA_DEV u32 coder_uint(RangeEnc &e, u32 val, u32 ft) { ec_enc_uint(e, val, ft); return val; }
CA_DEV u32 coder_uint(RangeDec &d, u32, u32 ft) { return ec_dec_uint(d, ft); }

}  // namespace ca

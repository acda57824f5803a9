// rangecoder.h -- the Opus range encoder (ec_enc) as wave-uniform device code.
//
// Counterpart of opus-fix/celt/entenc.c:62-508, entcode.c:69-96 and entcode.h:63-160. The coder state
// (rng/val/rem/ext/...) is held identically in the registers of every lane of the frame's wavefront;
// only lane 0 touches the output bytes (in LDS). All branches are wave-uniform.
// The tree-specific EC_DIFF debug field (entcode.h:92-93) has no device counterpart.
#pragma once
#include "celt_math.h"
#include "device_tables.h"

namespace ca {

enum { EC_SYM_BITS = 8, EC_CODE_BITS = 32, EC_SYM_MAX = 255, EC_CODE_SHIFT = 23, EC_UINT_BITS = 8, EC_WINDOW_SIZE = 32 };
#define CA_EC_CODE_TOP 0x80000000u
#define CA_EC_CODE_BOT 0x00800000u

struct RangeEnc {
    u8 *buf;            // LDS (or host memory in the emulation build)
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

CA_DEV void ec_enc_init(RangeEnc &e, u8 *buf, u32 size)               // entenc.c:163-184
{
    e.buf = buf;
    e.end_offs = 0;
    e.end_window = 0;
    e.nend_bits = 0;
    e.nbits_total = EC_CODE_BITS + 1;
    e.offs = 0;
    e.rng = CA_EC_CODE_TOP;
    e.rem = -1;
    e.val = 0;
    e.ext = 0;
    e.storage = size;
    e.error = 0;
}

CA_DEV int ec_tell(const RangeEnc &e) { return e.nbits_total - ec_ilog(e.rng); }   // entcode.h:114

CA_DEV u32 ec_tell_frac(const RangeEnc &e)                                           // entcode.c:69-96
{
    u32 nbits = (u32)e.nbits_total << 3;
    int l = ec_ilog(e.rng);
    u32 r = e.rng >> (l - 16);
    u32 b = (r >> 12) - 8;
    b += r > CLT_tell_frac_correction[b];
    return nbits - (u32)((l << 3) + (int)b);
}

CA_DEV int ec_write_byte(RangeEnc &e, u32 v)                                         // entenc.c:62-68
{
    if (e.offs + e.end_offs >= e.storage) return -1;
    if (lane() == 0) e.buf[e.offs] = (u8)v;
    e.offs++;
    return 0;
}

CA_DEV int ec_write_byte_at_end(RangeEnc &e, u32 v)                                  // entenc.c:70-100
{
    if (e.offs + e.end_offs >= e.storage) return -1;
    e.end_offs++;
    if (lane() == 0) e.buf[e.storage - e.end_offs] = (u8)v;
    return 0;
}

CA_DEV void ec_enc_carry_out(RangeEnc &e, int c)                                     // entenc.c:111-143
{
    if (c != EC_SYM_MAX) {
        int carry = c >> EC_SYM_BITS;
        if (e.rem >= 0) e.error |= ec_write_byte(e, (u32)(e.rem + carry));
        if (e.ext > 0) {
            u32 sym = (u32)(EC_SYM_MAX + carry) & EC_SYM_MAX;
            do e.error |= ec_write_byte(e, sym);
            while (--e.ext > 0);
        }
        e.rem = c & EC_SYM_MAX;
    } else {
        e.ext++;
    }
}

CA_DEV void ec_enc_normalize(RangeEnc &e)                                            // entenc.c:145-160
{
    while (e.rng <= CA_EC_CODE_BOT) {
        ec_enc_carry_out(e, (int)(e.val >> EC_CODE_SHIFT));
        e.val = (e.val << EC_SYM_BITS) & (CA_EC_CODE_TOP - 1);
        e.rng <<= EC_SYM_BITS;
        e.nbits_total += EC_SYM_BITS;
    }
}

CA_DEV void ec_encode(RangeEnc &e, u32 fl, u32 fh, u32 ft)                           // entenc.c:187-197
{
    CA_TRACE("1f 0x%x\n1g 0x%x\n1h 0x%x", fl, fh, ft);
    u32 r = e.rng / ft;
    if (fl > 0) {
        e.val += e.rng - r * (ft - fl);
        e.rng = r * (fh - fl);
    } else {
        e.rng -= r * (ft - fh);
    }
    ec_enc_normalize(e);
}

CA_DEV void ec_encode_bin(RangeEnc &e, u32 fl, u32 fh, u32 bits)                     // entenc.c:218-228
{
    CA_TRACE("1j 0x%x\n1k 0x%x", fh, bits);
    u32 r = e.rng >> bits;
    if (fl > 0) {
        e.val += e.rng - r * ((1u << bits) - fl);
        e.rng = r * (fh - fl);
    } else {
        e.rng -= r * ((1u << bits) - fh);
    }
    ec_enc_normalize(e);
}

CA_DEV void ec_enc_bit_logp(RangeEnc &e, int val, u32 logp)                          // entenc.c:249-261
{
    CA_TRACE("1l 0x%x\n1m 0x%x", (unsigned)val, logp);
    u32 r = e.rng, l = e.val, s = r >> logp;
    r -= s;
    if (val) e.val = l + r;
    e.rng = val ? s : r;
    ec_enc_normalize(e);
}

CA_DEV void ec_enc_icdf(RangeEnc &e, int s, const u8 *icdf, u32 ftb)                 // entenc.c:279-292
{
    CA_TRACE("1n 0x%x\n1p 0x%x", (unsigned)s, ftb);
    u32 r = e.rng >> ftb;
    if (s > 0) {
        e.val += e.rng - r * icdf[s - 1];
        e.rng = r * (u32)(icdf[s - 1] - icdf[s]);
    } else {
        e.rng -= r * icdf[s];
    }
    ec_enc_normalize(e);
}

CA_DEV void ec_enc_bits(RangeEnc &e, u32 fl, u32 bits)                               // entenc.c:346-365
{
    CA_TRACE("1s 0x%x\n1t 0x%x", fl, bits);
    u32 window = e.end_window;
    int used = e.nend_bits;
    if (used + (int)bits > EC_WINDOW_SIZE) {
        do {
            e.error |= ec_write_byte_at_end(e, window & EC_SYM_MAX);
            window >>= EC_SYM_BITS;
            used -= EC_SYM_BITS;
        } while (used >= EC_SYM_BITS);
    }
    window |= fl << used;
    used += (int)bits;
    e.end_window = window;
    e.nend_bits = used;
    e.nbits_total += (int)bits;
}

CA_DEV void ec_enc_uint(RangeEnc &e, u32 fl, u32 ft)                                 // entenc.c:313-329
{
    CA_TRACE("1q 0x%x\n1r 0x%x", fl, ft);
    ft--;
    int ftb = ec_ilog(ft);
    if (ftb > EC_UINT_BITS) {
        ftb -= EC_UINT_BITS;
        u32 ft1 = (ft >> ftb) + 1;
        u32 fl1 = fl >> ftb;
        ec_encode(e, fl1, fl1 + 1, ft1);
        ec_enc_bits(e, fl & ((1u << ftb) - 1u), (u32)ftb);
    } else {
        ec_encode(e, fl, fl + 1, ft + 1);
    }
}

CA_DEV void ec_enc_patch_initial_bits(RangeEnc &e, u32 val, u32 nbits)               // entenc.c:386-404
{
    int shift = EC_SYM_BITS - (int)nbits;
    u32 mask = ((1u << nbits) - 1) << shift;
    if (e.offs > 0) {
        if (lane() == 0) e.buf[0] = (u8)((e.buf[0] & ~mask) | val << shift);
    } else if (e.rem >= 0) {
        e.rem = (int)(((u32)e.rem & ~mask) | val << shift);
    } else if (e.rng <= (CA_EC_CODE_TOP >> nbits)) {
        e.val = (e.val & ~(mask << EC_CODE_SHIFT)) | val << (EC_CODE_SHIFT + shift);
    } else {
        e.error = -1;
    }
}

// Moves the raw-bit tail so the buffer ends at `size` (entenc.c:427-439). Caller must wave_sync()
// before any lane reads the moved bytes.
CA_DEV void ec_enc_shrink(RangeEnc &e, u32 size)
{
    if (lane() == 0 && e.end_offs > 0 && size != e.storage) {
        // regions may overlap; size < storage so moving ascending is safe (dst < src)
        u8 *dst = e.buf + size - e.end_offs;
        const u8 *src = e.buf + e.storage - e.end_offs;
        for (u32 i = 0; i < e.end_offs; i++) dst[i] = src[i];
    }
    e.storage = size;
}

CA_DEV void ec_enc_done(RangeEnc &e)                                                 // entenc.c:447-508
{
    int l = EC_CODE_BITS - ec_ilog(e.rng);
    u32 msk = (CA_EC_CODE_TOP - 1) >> l;
    u32 end = (e.val + msk) & ~msk;
    if ((end | msk) >= e.val + e.rng) {
        l++;
        msk >>= 1;
        end = (e.val + msk) & ~msk;
    }
    while (l > 0) {
        ec_enc_carry_out(e, (int)(end >> EC_CODE_SHIFT));
        end = (end << EC_SYM_BITS) & (CA_EC_CODE_TOP - 1);
        l -= EC_SYM_BITS;
    }
    if (e.rem >= 0 || e.ext > 0) ec_enc_carry_out(e, 0);
    u32 window = e.end_window;
    int used = e.nend_bits;
    while (used >= EC_SYM_BITS) {
        e.error |= ec_write_byte_at_end(e, window & EC_SYM_MAX);
        window >>= EC_SYM_BITS;
        used -= EC_SYM_BITS;
    }
    if (!e.error) {
        wave_sync();
        // zero the gap between the range-coded front and the raw-bit tail (all lanes help)
        int gap = (int)(e.storage - e.offs - e.end_offs);
        for (int i = lane(); i < gap; i += LANES) e.buf[e.offs + i] = 0;
        wave_sync();
        if (used > 0) {
            if (e.end_offs >= e.storage) {
                e.error = -1;
            } else {
                l = -l;
                if (e.offs + e.end_offs >= e.storage && l < used) {
                    window &= (1u << l) - 1;
                    e.error = -1;
                }
                if (lane() == 0) e.buf[e.storage - e.end_offs - 1] |= (u8)window;
            }
        }
    }
    wave_sync();
}

// ec_laplace_encode (celt/laplace.c:38-92). `value` may be clamped, as in the reference.
CA_DEV void ec_laplace_encode(RangeEnc &e, int &value, u32 fs, int decay)
{
    u32 fl = 0;
    int val = value;
    if (val) {
        int s = -(val < 0);
        val = (val + s) ^ s;
        fl = fs;
        fs = ((32768 - 32 - fs) * (u32)(16384 - decay)) >> 15;       // ec_laplace_get_freq1
        int i;
        for (i = 1; fs > 0 && i < val; i++) {
            fs *= 2;
            fl += fs + 2;
            fs = (fs * (u32)decay) >> 15;
        }
        if (!fs) {
            int ndi_max = (int)(32768 - fl + 1 - 1) >> 0;
            ndi_max = (ndi_max - s) >> 1;
            int di = imin(val - i, ndi_max - 1);
            fl += (u32)(2 * di + 1 + s);
            fs = (32768 - fl) < 1u ? (32768 - fl) : 1u;          // IMIN(LAPLACE_MINP, 32768-fl), unsigned
            value = (i + di + s) ^ s;
        } else {
            fs += 1;
            fl += fs & (u32)~s;
        }
    }
    ec_encode_bin(e, fl, fl + fs, 15);
}

}  // namespace ca

"""Range-coder scripts for tests/test_ec_script_{cpu,gpu}.py -- TEST INFRASTRUCTURE.

A script is a list of (opcode, a, b, c) rows (concentus_amd/csrc/ec_script.h). `run_reference` executes it with the compiled
reference's own ec_enc_* / ec_dec_* / ec_laplace_* functions (oracle/_ref/libopus_ref.so, celt/entenc.c, celt/entdec.c,
celt/laplace.c) on a reflib.EcCtx; the runners under test must leave the same ec_ctx fields, the same buffer bytes and (decoding)
return the same values. The cases restate what celt/tests/test_unit_entropy.c exercises: the raw patch_initial_bits vectors
(:325-358), the raw-bits overfill (:359-369), uniform integers over many ft (:71-108), random streams through all four symbol
methods (:148-259), plus Laplace symbols (celt/tests/test_unit_laplace.c)."""
import ctypes as C

import numpy as np

import reflib

ENC, ENC_BIN, BIT_LOGP, UINT, BITS, PATCH, SHRINK, DONE, LAPLACE, ICDF = range(10)
D_DECODE, D_DECODE_BIN, D_UPDATE, D_BIT_LOGP, D_UINT, D_BITS, D_LAPLACE, D_TELL, D_TELL_FRAC, D_ICDF = range(16, 26)
ICDF_TABLES = {0: (bytes([126, 124, 119, 109, 87, 41, 19, 9, 4, 2, 0]), 11), 1: (bytes([25, 23, 2, 0]), 4), 2: (bytes([2, 1, 0]), 3),
               3: (bytes([2, 1, 0]), 3)}
FIELDS = ("storage", "end_offs", "end_window", "nend_bits", "nbits_total", "offs", "rng", "val", "ext", "rem", "error")


def fresh_enc(size):
    """(EcCtx, buffer) as after ec_enc_init(&enc, buf, size) of the reference."""
    buf = np.zeros(max(size, 1), np.uint8)
    e = reflib.EcCtx()
    reflib.lib().ec_enc_init(C.byref(e), buf.ctypes.data_as(C.c_void_p), C.c_uint32(size))
    return e, buf


def fresh_dec(data):
    buf = np.ascontiguousarray(data, np.uint8).copy()
    e = reflib.EcCtx()
    reflib.lib().ec_dec_init(C.byref(e), buf.ctypes.data_as(C.c_void_p), C.c_uint32(len(buf)))
    return e, buf


def pack(e):
    return np.array([getattr(e, f) for f in FIELDS], dtype=np.int64).astype(np.uint32).view(np.int32).copy()


def unpack(e, v):
    for f, x in zip(FIELDS, np.asarray(v, np.int32)):
        x = int(x)
        setattr(e, f, x & 0xffffffff if f in ("storage", "end_offs", "end_window", "offs", "rng", "val", "ext") else x)


def run_reference(e, script):
    """Runs the script with the reference's functions on e (its buffer already attached); returns the decoder values."""
    r = reflib.lib()
    r.ec_decode.restype = r.ec_decode_bin.restype = r.ec_dec_uint.restype = r.ec_dec_bits.restype = r.ec_tell_frac.restype = C.c_uint32
    out = []
    u = C.c_uint32
    for row in script:
        op, a, b, c = (tuple(row) + (0, 0, 0))[:4]
        v = 0
        if op == ENC: r.ec_encode(C.byref(e), u(a), u(b), u(c))
        elif op == ENC_BIN: r.ec_encode_bin(C.byref(e), u(a), u(b), u(c))
        elif op == BIT_LOGP: r.ec_enc_bit_logp(C.byref(e), a, u(b))
        elif op == UINT: r.ec_enc_uint(C.byref(e), u(a), u(b))
        elif op == BITS: r.ec_enc_bits(C.byref(e), u(a), u(b))
        elif op == PATCH: r.ec_enc_patch_initial_bits(C.byref(e), u(a), u(b))
        elif op == SHRINK: r.ec_enc_shrink(C.byref(e), u(a))
        elif op == DONE: r.ec_enc_done(C.byref(e))
        elif op == LAPLACE:
            val = C.c_int(a)
            r.ec_laplace_encode(C.byref(e), C.byref(val), u(b), c)
        elif op == ICDF: r.ec_enc_icdf(C.byref(e), a, ICDF_TABLES[b][0], u(c))
        elif op == D_DECODE: v = r.ec_decode(C.byref(e), u(a))
        elif op == D_DECODE_BIN: v = r.ec_decode_bin(C.byref(e), u(a))
        elif op == D_UPDATE: r.ec_dec_update(C.byref(e), u(a), u(b), u(c))
        elif op == D_BIT_LOGP: v = r.ec_dec_bit_logp(C.byref(e), u(a))
        elif op == D_UINT: v = r.ec_dec_uint(C.byref(e), u(a))
        elif op == D_BITS: v = r.ec_dec_bits(C.byref(e), u(a))
        elif op == D_LAPLACE: v = r.ec_laplace_decode(C.byref(e), u(a), b)
        elif op == D_TELL: v = e.nbits_total - (32 - _clz32(e.rng))
        elif op == D_TELL_FRAC: v = r.ec_tell_frac(C.byref(e))
        elif op == D_ICDF: v = r.ec_dec_icdf(C.byref(e), ICDF_TABLES[a][0], u(b))
        else: raise ValueError(op)
        out.append(int(v) & 0xffffffff)
    return np.array(out, dtype=np.int64).astype(np.uint32).view(np.int32)


def _clz32(v):
    return 32 - int(v).bit_length()


def ops_array(script):
    a = np.zeros((len(script), 4), np.int64)
    for k, row in enumerate(script):
        a[k, :len(row)] = row
    return np.ascontiguousarray(a.astype(np.uint32).view(np.int32).reshape(len(script), 4))


def known_answer_cases():
    """(name, buffer size, script, expected) from celt/tests/test_unit_entropy.c:325-369; expected = dict of checks."""
    return [
        # :325-343: five bits, patch two leading bits to 3 -> ok; patching five bits of a coder whose range has shrunk -> error;
        # one byte, value 192
        ("patch 3/2 then 0/5", 1275, [(BIT_LOGP, 0, 1), (BIT_LOGP, 0, 1), (BIT_LOGP, 0, 1), (BIT_LOGP, 0, 1), (BIT_LOGP, 0, 2), (PATCH, 3, 2)],
         {"error": 0}),
        ("patch 3/2 then 0/5, second", 1275, [(BIT_LOGP, 0, 1), (BIT_LOGP, 0, 1), (BIT_LOGP, 0, 1), (BIT_LOGP, 0, 1), (BIT_LOGP, 0, 2), (PATCH, 3, 2),
                                              (PATCH, 0, 5), (DONE,)], {"error": -1, "range_bytes": 1, "byte0": 192}),
        # :344-358: two bytes, first 63
        ("patch 0/2 over a finished byte", 1275, [(BIT_LOGP, 0, 1), (BIT_LOGP, 0, 1), (BIT_LOGP, 1, 6), (BIT_LOGP, 0, 2), (PATCH, 0, 2), (DONE,)],
         {"error": 0, "range_bytes": 2, "byte0": 63}),
        # :359-369: 48 raw bits into a 2-byte buffer must fail
        ("raw bits overfill", 2, [(BIT_LOGP, 0, 2)] + [(BITS, 0, 1)] * 48 + [(DONE,)], {"error": -1}),
        # :370-381: 17 raw bits into 2 bytes after a range-coded bit: raw bits win, error set
        ("17 raw bits in 2 bytes", 2, [(BITS, 0x55, 7)] + [(BITS, 0, 1)] * 10 + [(BIT_LOGP, 0, 2), (DONE,)], {"error": -1}),
    ]


def random_stream_case(seed, n_syms=None):
    """One iteration of test_unit_entropy.c:148-259: sz symbols of ft = 2..1024, each through a random one of the four
    methods, then ec_enc_done; returns (buffer size, encode script, decode script)."""
    rng = np.random.default_rng(seed)
    ft = int(rng.integers(2, 1025))
    sz = n_syms or int(rng.integers(1, 513))
    data = rng.integers(0, ft, size=sz)
    zeros = int(rng.integers(0, 13)) == 0
    logp1 = [int(rng.integers(1, 17)) for _ in range(sz)]
    enc, dec = [], []
    for j in range(sz):
        if zeros:
            data[j] = 0
        d = int(data[j])
        m = int(rng.integers(0, 4))
        if m == 0:
            enc.append((ENC, d, d + 1, ft))
        elif m == 1:
            bits = max(1, (ft - 1).bit_length())
            enc.append((ENC_BIN, d, d + 1, bits))
        elif m == 2:
            enc.append((BIT_LOGP, d & 1, logp1[j]))
        else:
            t = int(rng.integers(0, 4))
            n = ICDF_TABLES[t][1]
            ftb = 7 if t == 0 else 5 if t == 1 else 2
            enc.append((ICDF, d % n, t, ftb))
        dm = int(rng.integers(0, 2))
        if m == 0:
            dec += [(D_DECODE, ft), (D_UPDATE, d, d + 1, ft)] if dm == 0 else [(D_DECODE, ft), (D_UPDATE, d, d + 1, ft)]
        elif m == 1:
            bits = max(1, (ft - 1).bit_length())
            dec += [(D_DECODE_BIN, bits), (D_UPDATE, d, d + 1, 1 << bits)]
        elif m == 2:
            dec.append((D_BIT_LOGP, logp1[j]))
        else:
            dec.append((D_ICDF, enc[-1][2], enc[-1][3]))
        dec.append((D_TELL_FRAC,))
    enc.append((DONE,))
    return 1275, enc, dec


def uint_bits_case(seed, n=250):
    """test_unit_entropy.c:71-147 in miniature: ec_enc_uint over many ft and ec_enc_bits over many widths, interleaved, plus
    Laplace symbols; decode script mirrors it."""
    rng = np.random.default_rng(seed)
    enc, dec = [], []
    for _ in range(n):
        k = int(rng.integers(0, 3))
        if k == 0:
            ft = int(rng.integers(2, 1 << int(rng.integers(2, 31))))
            v = int(rng.integers(0, ft))
            enc.append((UINT, v, ft)); dec.append((D_UINT, ft))
        elif k == 1:
            b = int(rng.integers(1, 26))
            v = int(rng.integers(0, 1 << b))
            enc.append((BITS, v, b)); dec.append((D_BITS, b))
        else:
            fs = int(rng.integers(1, 32000))
            decay = int(rng.integers(0, 11456))
            v = int(rng.integers(-30, 31))
            enc.append((LAPLACE, v, fs, decay)); dec.append((D_LAPLACE, fs, decay))
        dec.append((D_TELL,))
    enc.append((DONE,))
    return 1275, enc, dec

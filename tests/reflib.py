"""ctypes access to the compiled reference (oracle/_ref/libopus_ref.so) -- TEST INFRASTRUCTURE.

The library is the unmodified opus-fix tree built by oracle/Makefile (FIXED_POINT). Struct
layouts mirror opus-fix/celt/modes.h:52-76, celt/mdct.h:49-54, celt/kiss_fft.h:75-86 and
celt/entcode.h:63-94 (x86-64 SysV ABI). Nothing here is imported by the product package.
"""
import ctypes as C
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_PATH = os.path.join(ROOT, "oracle", "_ref", "libopus_ref.so")


class KissFFTState(C.Structure):
    _fields_ = [
        ("nfft", C.c_int),
        ("scale", C.c_int16),
        ("scale_shift", C.c_int),
        ("shift", C.c_int),
        ("factors", C.c_int16 * 16),
        ("bitrev", C.POINTER(C.c_int16)),
        ("twiddles", C.POINTER(C.c_int16)),
        ("arch_fft", C.c_void_p),
    ]


class MdctLookup(C.Structure):
    _fields_ = [
        ("n", C.c_int),
        ("maxshift", C.c_int),
        ("kfft", C.POINTER(KissFFTState) * 4),
        ("trig", C.POINTER(C.c_int16)),
    ]


class PulseCache(C.Structure):
    _fields_ = [
        ("size", C.c_int),
        ("index", C.POINTER(C.c_int16)),
        ("bits", C.POINTER(C.c_uint8)),
        ("caps", C.POINTER(C.c_uint8)),
    ]


class CELTMode(C.Structure):
    _fields_ = [
        ("Fs", C.c_int32),
        ("overlap", C.c_int),
        ("nbEBands", C.c_int),
        ("effEBands", C.c_int),
        ("preemph", C.c_int16 * 4),
        ("eBands", C.POINTER(C.c_int16)),
        ("maxLM", C.c_int),
        ("nbShortMdcts", C.c_int),
        ("shortMdctSize", C.c_int),
        ("nbAllocVectors", C.c_int),
        ("allocVectors", C.POINTER(C.c_uint8)),
        ("logN", C.POINTER(C.c_int16)),
        ("window", C.POINTER(C.c_int16)),
        ("mdct", MdctLookup),
        ("cache", PulseCache),
    ]


class EcCtx(C.Structure):
    """ec_ctx incl. the tree-specific trailing EC_DIFF field (celt/entcode.h:92-93)."""
    _fields_ = [
        ("buf", C.POINTER(C.c_ubyte)),
        ("storage", C.c_uint32),
        ("end_offs", C.c_uint32),
        ("end_window", C.c_uint32),
        ("nend_bits", C.c_int),
        ("nbits_total", C.c_int),
        ("offs", C.c_uint32),
        ("rng", C.c_uint32),
        ("val", C.c_uint32),
        ("ext", C.c_uint32),
        ("rem", C.c_int),
        ("error", C.c_int),
        ("EC_DIFF", C.c_int),
    ]


_lib = None


def available():
    return os.path.exists(REF_PATH)


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(REF_PATH)
        _lib.opus_custom_mode_create.restype = C.POINTER(CELTMode)
        _lib.opus_custom_mode_create.argtypes = [C.c_int32, C.c_int, C.POINTER(C.c_int)]
        _lib.opus_encoder_create.restype = C.c_void_p
        _lib.opus_encoder_create.argtypes = [C.c_int32, C.c_int, C.c_int, C.POINTER(C.c_int)]
        _lib.opus_encoder_destroy.argtypes = [C.c_void_p]
        _lib.opus_encode.restype = C.c_int32
        _lib.opus_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int32]
        _lib.opus_decoder_create.restype = C.c_void_p
        _lib.opus_decoder_create.argtypes = [C.c_int32, C.c_int, C.POINTER(C.c_int)]
        _lib.opus_decoder_destroy.argtypes = [C.c_void_p]
        _lib.opus_decode.restype = C.c_int
        _lib.opus_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int, C.c_int]
    return _lib


def mode():
    err = C.c_int(0)
    m = lib().opus_custom_mode_create(48000, 960, C.byref(err))
    assert err.value == 0
    return m.contents


def arr(ptr, n):
    return [ptr[i] for i in range(n)]


_syms = None


def static_table(name, fmt, count):
    """Read a file-local `static const` array of the reference out of the .so's .rodata
    (symbol address from `nm`; for .rodata the virtual address equals the file offset here)."""
    import struct
    import subprocess
    global _syms
    if _syms is None:
        _syms = {}
        for line in subprocess.check_output(["nm", REF_PATH], text=True).splitlines():
            parts = line.split()
            if len(parts) == 3 and parts[1] in "rR":
                _syms[parts[2].split(".")[0]] = int(parts[0], 16)
    off = _syms[name]
    size = struct.calcsize("<" + fmt) * count
    with open(REF_PATH, "rb") as f:
        f.seek(off)
        raw = f.read(size)
    return list(struct.unpack("<%d%s" % (count, fmt), raw))

"""Host-side mirror of the reference's MDCT operator interface (opus-fix/celt/mdct.h:56-110).

`clt_mdct_forward` / `clt_mdct_backward` keep the reference's argument meaning (one transform, host
arrays, `shift`, `stride`); `mdct_forward_batch` / `mdct_backward_batch` are the throughput path over
device-resident tensors laid out as compute_mdcts / celt_synthesis lay out one frame
(celt/celt_encoder.c:418-461, celt/celt_decoder.c:323-346).
"""
import ctypes as C

import numpy as np

from . import lib as _lib

OVERLAP = 120
FRAME = 960


class _MdctLookupHead(C.Structure):
    # leading fields of mdct_lookup (celt/mdct.h:49-54) -- all the device path validates
    _fields_ = [("n", C.c_int), ("maxshift", C.c_int)]


_STATIC_LOOKUP = _MdctLookupHead(1920, 3)
_WINDOW_TOKEN = (C.c_int16 * 1)()


def _np32(a, n, name):
    a = np.ascontiguousarray(a, dtype=np.int32)
    if a.size < n:
        raise ValueError("%s: need at least %d int32 values, got %d" % (name, n, a.size))
    return a


def clt_mdct_forward(x, shift=0, stride=1, out=None):
    """One forward MDCT (N = 1920 >> shift) of x[N/2 + 120]; returns out with coefficient k at
    out[k*stride]. Mirrors clt_mdct_forward(l, in, out, window, overlap, shift, stride, arch)."""
    n2 = FRAME >> shift
    x = _np32(x, n2 + OVERLAP, "x").copy()
    if out is None:
        out = np.zeros((n2 - 1) * stride + 1, np.int32)
    L = _lib.load()
    L.opusgpu_clt_mdct_forward(C.byref(_STATIC_LOOKUP), x.ctypes.data, out.ctypes.data,
                               C.addressof(_WINDOW_TOKEN), OVERLAP, shift, stride, 0)
    _lib.check(L.opusgpu_get_last_error(), "opusgpu_clt_mdct_forward")
    return out


def clt_mdct_backward(coef, out, shift=0, stride=1):
    """One inverse MDCT: coefficient k read from coef[k*stride]; `out` (N/2 + 120 int32) holds the
    previous tail in out[0:60] on entry and is updated in place, as the reference does."""
    n2 = FRAME >> shift
    coef = _np32(coef, (n2 - 1) * stride + 1, "coef")
    if out.dtype != np.int32 or out.size < n2 + OVERLAP or not out.flags.c_contiguous:
        raise ValueError("out must be a contiguous int32 array of at least N/2+120 samples")
    L = _lib.load()
    L.opusgpu_clt_mdct_backward(C.byref(_STATIC_LOOKUP), coef.ctypes.data, out.ctypes.data,
                                C.addressof(_WINDOW_TOKEN), OVERLAP, shift, stride, 0)
    _lib.check(L.opusgpu_get_last_error(), "opusgpu_clt_mdct_backward")
    return out


def _check_dev(t, shape_tail, name):
    import torch
    if not (t.is_cuda and t.dtype == torch.int32 and t.is_contiguous()):
        raise ValueError("%s must be a contiguous int32 CUDA tensor" % name)
    if t.dim() != 3 or t.shape[2] != shape_tail or t.shape[1] not in (1, 2):
        raise ValueError("%s must have shape [frames][channels<=2][%d]" % (name, shape_tail))


def mdct_forward_batch(sig, freq=None, shift=0):
    """sig int32 [F][C][1080] (device) -> freq int32 [F][C][960] (device). Asynchronous on the
    current torch stream."""
    import torch
    _check_dev(sig, 1080, "sig")
    if freq is None:
        freq = torch.empty((sig.shape[0], sig.shape[1], FRAME), dtype=torch.int32, device=sig.device)
    _check_dev(freq, FRAME, "freq")
    rc = _lib.load().opusgpu_mdct_forward_batch(sig.data_ptr(), freq.data_ptr(), sig.shape[0], sig.shape[1],
                                                shift, _lib.current_stream_handle())
    _lib.check(rc, "opusgpu_mdct_forward_batch")
    return freq


def mdct_backward_batch(freq, sig, shift=0):
    """freq int32 [F][C][960] + sig int32 [F][C][1080] (sig[..., :60] = previous tail) -> sig updated
    in place ([..., :1020] written)."""
    _check_dev(freq, FRAME, "freq")
    _check_dev(sig, 1080, "sig")
    if freq.shape[:2] != sig.shape[:2]:
        raise ValueError("freq and sig disagree on [frames][channels]")
    rc = _lib.load().opusgpu_mdct_backward_batch(freq.data_ptr(), sig.data_ptr(), sig.shape[0], sig.shape[1],
                                                 shift, _lib.current_stream_handle())
    _lib.check(rc, "opusgpu_mdct_backward_batch")
    return sig


class _KissFftStateHead(C.Structure):
    # leading fields of kiss_fft_state (celt/kiss_fft.h:75-85, FIXED_POINT) -- all the device path validates
    _fields_ = [("nfft", C.c_int), ("scale", C.c_int16), ("scale_shift", C.c_int), ("shift", C.c_int)]


def opus_fft(fin, shift=0, cfg=None):
    """opus_fft(cfg, fin, fout) for one transform of 480 >> shift complex points: fin int32 [nfft][2] (host) -> fout.
    `cfg`: a pointer to one of the static mode's kiss_fft_state objects; by default a head with the same leading fields."""
    nfft = 480 >> shift
    fin = _np32(fin, 2 * nfft, "fin")
    fout = np.zeros(2 * nfft, np.int32)
    head = _KissFftStateHead(nfft, 17476, 8 - shift, shift if shift else -1)
    L = _lib.load()
    L.opusgpu_opus_fft(cfg if cfg is not None else C.byref(head), fin.ctypes.data, fout.ctypes.data)
    _lib.check(L.opusgpu_get_last_error(), "opusgpu_opus_fft")
    return fout.reshape(nfft, 2)


def fft_batch(fin, shift=0, fout=None):
    """Batched opus_fft: fin int32 CUDA tensor [n][480 >> shift][2] -> fout of the same shape (out of place)."""
    import torch
    nfft = 480 >> shift
    if not (fin.is_cuda and fin.dtype == torch.int32 and fin.is_contiguous() and fin.dim() == 3 and tuple(fin.shape[1:]) == (nfft, 2)):
        raise ValueError("fin must be a contiguous int32 CUDA tensor [n][%d][2]" % nfft)
    if fout is None:
        fout = torch.empty_like(fin)
    _lib.check(_lib.load().opusgpu_fft_batch(fin.data_ptr(), fout.data_ptr(), fin.shape[0], shift, _lib.current_stream_handle()),
               "opusgpu_fft_batch")
    return fout

"""Host-side mirror of the reference's encoder API for the CELT-only hot path.

The reference drives one stream at a time: opus_encoder_create() -> opus_encoder_ctl(...) ->
opus_encode() per 20 ms frame (opus-fix/include/opus.h:164-263, src/opus_encoder.c:482,2031,2007).
`OpusEncoderBatch` keeps the same three verbs -- create / ctl / encode -- over N streams whose state
lives in HBM, and `encode_independent` is the stateless form used for BASELINE config #3 (each frame
is the first frame of its own stream). All computation happens in libopusgpu.so.
"""
import ctypes as C
import os

from . import lib as _lib

OPUS_APPLICATION_RESTRICTED_LOWDELAY = 2051
OPUS_AUTO = -1000
OPUS_BITRATE_MAX = -1
# ctl requests (opus-fix/include/opus_defines.h:130-167)
OPUS_SET_BITRATE_REQUEST = 4002
OPUS_SET_VBR_REQUEST = 4006
OPUS_SET_BANDWIDTH_REQUEST = 4008
OPUS_SET_COMPLEXITY_REQUEST = 4010
OPUS_SET_INBAND_FEC_REQUEST = 4012
OPUS_SET_PACKET_LOSS_PERC_REQUEST = 4014
OPUS_SET_DTX_REQUEST = 4016
OPUS_SET_VBR_CONSTRAINT_REQUEST = 4020
OPUS_SET_FORCE_CHANNELS_REQUEST = 4022
OPUS_GET_FINAL_RANGE_REQUEST = 4031
OPUS_SET_LSB_DEPTH_REQUEST = 4036
OPUS_SET_EXPERT_FRAME_DURATION_REQUEST = 4040
OPUS_FRAMESIZE_ARG = 5000
FRAME_SIZE = 960


class CeltConfig(C.Structure):
    """opusgpu_celt_config (include/opusgpu.h)."""
    _fields_ = [(n, C.c_int32) for n in
                "channels bitrate vbr constrained_vbr complexity lsb_depth loss_rate max_data_bytes".split()]


def default_config(channels=2, bitrate=96000):
    """What opus_encoder_create(48000, ch, RESTRICTED_LOWDELAY) + opus_demo's ctl sequence yields
    (src/opus_demo.c:531-543: VBR on, unconstrained, complexity 10, 16-bit, 1500-byte buffer)."""
    return CeltConfig(channels, bitrate, 1, 0, 10, 16, 0, 1500)


def out_stride_for(cfg):
    return (min(cfg.max_data_bytes, 1276) + 3) & ~3


def _check_pcm(pcm, channels):
    import torch
    if not (pcm.is_cuda and pcm.dtype == torch.int16 and pcm.is_contiguous()):
        raise ValueError("pcm must be a contiguous int16 CUDA tensor")
    if pcm.dim() != 3 or pcm.shape[1] != FRAME_SIZE or pcm.shape[2] != channels:
        raise ValueError("pcm must have shape [frames][960][%d]" % channels)


_WORKSPACE = {}
MID_RECORD_BYTES = 4704         # sizeof(FrameMid) in csrc/celt_enc.h (hand-off record per frame)
WORKSPACE_FRAMES = int(os.environ.get("CONCENTUS_WS_FRAMES", 262144))      # hand-off records kept in HBM at once (larger batches are chunked by the library)


def _workspace(device, n_frames):
    """Scratch for the hand-off between the kernels of one batch, per (device, stream): allocated once and
    reused; batches launched on different streams (double-buffered callers) must not share it."""
    import torch
    L = _lib.load()
    need = L.opusgpu_encode_workspace_bytes(min(max(n_frames, 1), WORKSPACE_FRAMES))
    key = (device, _lib.current_stream_handle().value)
    ws = _WORKSPACE.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty((need,), dtype=torch.uint8, device=device)
        _WORKSPACE[key] = ws
    return ws


def _encode(cfg, states_ptr, pcm):
    import torch
    _check_pcm(pcm, cfg.channels)
    n = pcm.shape[0]
    stride = out_stride_for(cfg)
    out = torch.zeros((n, stride), dtype=torch.uint8, device=pcm.device)
    lens = torch.empty((n,), dtype=torch.int32, device=pcm.device)
    rng = torch.empty((n,), dtype=torch.int32, device=pcm.device)     # uint32 bit pattern
    ws = _workspace(pcm.device, n)
    rc = _lib.load().opusgpu_encode_batch(C.byref(cfg), states_ptr, pcm.data_ptr(), out.data_ptr(), stride,
                                          lens.data_ptr(), rng.data_ptr(), n, ws.data_ptr(), ws.numel(),
                                          _lib.current_stream_handle())
    _lib.check(rc, "opusgpu_encode_batch")
    return out, lens, rng


def encode_independent(pcm, cfg=None):
    """Encode every row of pcm [F][960][C] as the first frame of its own fresh stream.
    Returns (packets uint8 [F][stride], lengths int32 [F], final_range int32-bit-pattern [F])."""
    cfg = cfg or default_config(pcm.shape[2])
    return _encode(cfg, None, pcm)


class OpusEncoderBatch:
    """N encoders created alike; encode() advances every stream by one 20 ms frame."""

    def __init__(self, n_streams, Fs=48000, channels=2, application=OPUS_APPLICATION_RESTRICTED_LOWDELAY, device="cuda"):
        import torch
        if Fs != 48000 or application != OPUS_APPLICATION_RESTRICTED_LOWDELAY:
            raise _lib.OpusGpuError(-5, "opus_encoder_create: only 48 kHz RESTRICTED_LOWDELAY is implemented")
        if channels not in (1, 2):
            raise _lib.OpusGpuError(-1, "opus_encoder_create")
        self.n = n_streams
        # opus_encoder_create defaults (src/opus_encoder.c:164-252): auto bitrate, VBR on, constrained, complexity 9
        self.cfg = CeltConfig(channels, 60 * 50 + 48000 * channels, 1, 1, 9, 24, 0, 1500)
        L = _lib.load()
        self._states = torch.empty((n_streams, L.opusgpu_celt_state_size()), dtype=torch.uint8, device=device)
        _lib.check(L.opusgpu_celt_state_init(self._states.data_ptr(), n_streams, _lib.current_stream_handle()),
                   "opusgpu_celt_state_init")
        self.final_range = None
        self._bitrate_is_max = False

    def ctl(self, request, value=None):
        """opus_encoder_ctl(enc, request, value) for the requests opus_demo issues."""
        c = self.cfg
        if request == OPUS_SET_BITRATE_REQUEST:
            if value == OPUS_AUTO:
                value = 60 * 50 + 48000 * c.channels
            if value != OPUS_BITRATE_MAX and value <= 0:
                raise _lib.OpusGpuError(-1, "OPUS_SET_BITRATE")
            # OPUS_BITRATE_MAX is resolved per encode() call from that call's buffer size (user_bitrate_to_bitrate,
            # src/opus_encoder.c:512-521)
            self._bitrate_is_max = value == OPUS_BITRATE_MAX
            c.bitrate = max(500, min(value, 300000 * c.channels)) if value != OPUS_BITRATE_MAX else 1276 * 400
        elif request == OPUS_SET_VBR_REQUEST:
            c.vbr = int(bool(value))
        elif request == OPUS_SET_VBR_CONSTRAINT_REQUEST:
            c.constrained_vbr = int(bool(value))
        elif request == OPUS_SET_COMPLEXITY_REQUEST:
            if not 0 <= value <= 10:
                raise _lib.OpusGpuError(-1, "OPUS_SET_COMPLEXITY")
            c.complexity = value
        elif request == OPUS_SET_PACKET_LOSS_PERC_REQUEST:
            if not 0 <= value <= 100:
                raise _lib.OpusGpuError(-1, "OPUS_SET_PACKET_LOSS_PERC")
            c.loss_rate = value
        elif request == OPUS_SET_LSB_DEPTH_REQUEST:
            if not 8 <= value <= 24:
                raise _lib.OpusGpuError(-1, "OPUS_SET_LSB_DEPTH")
            c.lsb_depth = min(value, 16)       # opus_encode() passes 16 for int16 input (src/opus_encoder.c:2022)
        elif request in (OPUS_SET_BANDWIDTH_REQUEST, OPUS_SET_FORCE_CHANNELS_REQUEST):
            if value != OPUS_AUTO:
                raise _lib.OpusGpuError(-5, "only OPUS_AUTO is implemented for this request")
        elif request in (OPUS_SET_INBAND_FEC_REQUEST, OPUS_SET_DTX_REQUEST):
            if value:
                raise _lib.OpusGpuError(-5, "FEC/DTX are SILK features; not on the CELT-only path")
        elif request == OPUS_SET_EXPERT_FRAME_DURATION_REQUEST:
            if value != OPUS_FRAMESIZE_ARG:
                raise _lib.OpusGpuError(-5, "only OPUS_FRAMESIZE_ARG is implemented")
        elif request == OPUS_GET_FINAL_RANGE_REQUEST:
            return self.final_range
        else:
            raise _lib.OpusGpuError(-5, "opus_encoder_ctl request %d" % request)
        return 0

    def apply_opus_demo_ctls(self, bitrate=96000, vbr=1, cvbr=0, complexity=10):
        """The sequence at src/opus_demo.c:531-543."""
        for req, v in ((OPUS_SET_BITRATE_REQUEST, bitrate), (OPUS_SET_BANDWIDTH_REQUEST, OPUS_AUTO),
                       (OPUS_SET_VBR_REQUEST, vbr), (OPUS_SET_VBR_CONSTRAINT_REQUEST, cvbr),
                       (OPUS_SET_COMPLEXITY_REQUEST, complexity), (OPUS_SET_INBAND_FEC_REQUEST, 0),
                       (OPUS_SET_FORCE_CHANNELS_REQUEST, OPUS_AUTO), (OPUS_SET_DTX_REQUEST, 0),
                       (OPUS_SET_PACKET_LOSS_PERC_REQUEST, 0), (OPUS_SET_LSB_DEPTH_REQUEST, 16),
                       (OPUS_SET_EXPERT_FRAME_DURATION_REQUEST, OPUS_FRAMESIZE_ARG)):
            self.ctl(req, v)
        return self

    def encode(self, pcm, max_data_bytes=1500):
        """opus_encode(enc, pcm, 960, data, max_data_bytes) for every stream: pcm int16 [N][960][C]."""
        if pcm.shape[0] != self.n:
            raise ValueError("one frame per stream expected")
        self.cfg.max_data_bytes = max_data_bytes
        if self._bitrate_is_max:
            self.cfg.bitrate = min(1276, max_data_bytes) * 8 * (48000 // FRAME_SIZE)
        out, lens, rng = _encode(self.cfg, self._states.data_ptr(), pcm)
        self.final_range = rng
        return out, lens

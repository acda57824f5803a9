"""Host-side mirror of the reference's decoder API for the CELT-only path.

The reference decodes one stream at a time: opus_decoder_create() -> opus_decode() per packet
(opus-fix/include/opus.h:438-462, src/opus_decoder.c:121,758). `OpusDecoderBatch` keeps the same verbs over N
streams whose state (overlap/history buffer, band energies, post-filter) lives in HBM; all computation happens
in libopusgpu.so (opusgpu_decode_batch)."""
from . import lib as _lib

OPUS_GET_FINAL_RANGE_REQUEST = 4031


class OpusDecoderBatch:
    """N decoders created alike; decode() consumes the next packet of every stream."""

    def __init__(self, n_streams, Fs=48000, channels=2, device="cuda"):
        import torch
        if Fs != 48000 or channels != 2:
            raise _lib.OpusGpuError(-5, "opus_decoder_create: only 48 kHz stereo is implemented")
        self.n = n_streams
        L = _lib.load()
        self._states = torch.empty((n_streams, L.opusgpu_celt_dec_state_size()), dtype=torch.uint8, device=device)
        self.reset()
        self.final_range = None

    def reset(self):
        """OPUS_RESET_STATE for every stream."""
        _lib.check(_lib.load().opusgpu_celt_dec_state_init(self._states.data_ptr(), self.n, _lib.current_stream_handle()),
                   "opusgpu_celt_dec_state_init")

    def decode(self, packets, lengths):
        """opus_decode(dec, data, len, pcm, 960, 0) for every stream: packets uint8 [N][stride], lengths int32 [N].
        Returns (pcm int16 [N][960][2], ret int32 [N] = 960 or a negative error per stream)."""
        import torch
        if packets.dtype != torch.uint8 or packets.dim() != 2 or packets.shape[0] != self.n or not packets.is_contiguous():
            raise ValueError("packets must be a contiguous uint8 tensor [streams][stride]")
        if lengths.dtype != torch.int32 or lengths.shape != (self.n,):
            raise ValueError("lengths must be int32 [streams]")
        pcm = torch.empty((self.n, 960, 2), dtype=torch.int16, device=packets.device)
        ret = torch.empty((self.n,), dtype=torch.int32, device=packets.device)
        rng = torch.empty((self.n,), dtype=torch.int32, device=packets.device)
        rc = _lib.load().opusgpu_decode_batch(self._states.data_ptr(), packets.data_ptr(), packets.shape[1], lengths.data_ptr(),
                                              pcm.data_ptr(), ret.data_ptr(), rng.data_ptr(), self.n, _lib.current_stream_handle())
        _lib.check(rc, "opusgpu_decode_batch")
        self.final_range = rng
        return pcm, ret

    def ctl(self, request):
        if request == OPUS_GET_FINAL_RANGE_REQUEST:
            return self.final_range
        raise _lib.OpusGpuError(-5, "opus_decoder_ctl request %d" % request)


def decode_independent(packets, lengths):
    """Decode every packet with its own fresh decoder (the first packet of its own stream)."""
    dec = OpusDecoderBatch(packets.shape[0], device=packets.device)
    pcm, ret = dec.decode(packets, lengths)
    return pcm, ret, dec.final_range

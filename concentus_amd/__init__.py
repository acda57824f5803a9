"""concentus_amd -- MI355X (gfx950) batched Opus frame path behind a C-ABI (libopusgpu.so).

Host-side mirror of the reference's operator interface for the hot path only
(opus-fix: CELT MDCT, PVQ band quantiser, range coder, SILK Burg/NSQ). PyTorch is used for device
memory and streams; every computation happens in hand-written HIP kernels reached through
`include/opusgpu.h`. There is no CPU fallback: a missing library raises.
"""
from . import lib  # noqa: F401
from .mdct import clt_mdct_forward, clt_mdct_backward, mdct_forward_batch, mdct_backward_batch, opus_fft, fft_batch  # noqa: F401
from .encoder import OpusEncoderBatch, encode_independent, default_config, CeltConfig  # noqa: F401
from .silk import silk_burg_modified, silk_NSQ, silk_NSQ_del_dec, silk_find_LPC, silk_process_NLSFs, silk_residual_energy, silk_find_pred_coefs, silk_process_gains, silk_noise_shape_analysis, silk_prefilter, silk_find_pitch_lags, silk_encode_bits, silk_VAD_GetSA_Q8, silk_rate_control  # noqa: F401
from .silk_chain import SilkAnalysisChain  # noqa: F401
from .decoder import OpusDecoderBatch, decode_independent  # noqa: F401

__all__ = ["lib", "OpusDecoderBatch", "decode_independent", "clt_mdct_forward", "clt_mdct_backward", "mdct_forward_batch", "mdct_backward_batch", "opus_fft", "fft_batch",
           "OpusEncoderBatch", "encode_independent", "default_config", "CeltConfig",
           "silk_burg_modified", "silk_NSQ", "silk_NSQ_del_dec", "silk_find_LPC", "silk_process_NLSFs",
           "silk_residual_energy", "silk_find_pred_coefs", "silk_process_gains",
           "silk_noise_shape_analysis", "silk_prefilter", "silk_find_pitch_lags",
           "silk_encode_bits", "silk_VAD_GetSA_Q8", "silk_rate_control", "SilkAnalysisChain"]

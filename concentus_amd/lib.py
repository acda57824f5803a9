"""ctypes binding of libopusgpu.so (the C-ABI declared in include/opusgpu.h).

The library is built in-tree by `__graft_entry__.build()` / `make -C concentus_amd/csrc`.
Loading fails loudly: there is no fallback path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libopusgpu.so")

# every symbol include/opusgpu.h declares: (name, restype, argtypes)
_vp, _i = C.c_void_p, C.c_int
SYMBOLS = [
    ("opusgpu_get_version_string", C.c_char_p, []),
    ("opusgpu_strerror", C.c_char_p, [_i]),
    ("opusgpu_get_last_error", _i, []),
    ("opusgpu_num_cus", _i, []),
    ("opusgpu_kernel_timing_enable", _i, [_i]),
    ("opusgpu_kernel_timing_read", _i, [_vp, _vp, _i]),
    ("opusgpu_mdct_forward_batch", _i, [_vp, _vp, _i, _i, _i, _vp]),
    ("opusgpu_mdct_backward_batch", _i, [_vp, _vp, _i, _i, _i, _vp]),
    ("opusgpu_clt_mdct_forward", None, [_vp, _vp, _vp, _vp, _i, _i, _i, _i]),
    ("opusgpu_clt_mdct_backward", None, [_vp, _vp, _vp, _vp, _i, _i, _i, _i]),
    ("opusgpu_fft_batch", _i, [_vp, _vp, _i, _i, _vp]),
    ("opusgpu_opus_fft", None, [_vp, _vp, _vp]),
    ("opusgpu_celt_pitch_xcorr", C.c_int32, [_vp, _vp, _vp, _i, _i, _i]),
    ("opusgpu_celt_state_size", _i, []),
    ("opusgpu_celt_state_init", _i, [_vp, _i, _vp]),
    ("opusgpu_encode_workspace_bytes", C.c_size_t, [_i]),
    ("opusgpu_encode_batch", _i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp, C.c_size_t, _vp]),
    ("opusgpu_silk_burg_modified_batch", _i, [_vp, _vp, _i, _vp]),
    ("opusgpu_silk_burg_modified_c", None, [_vp, _vp, _vp, _vp, C.c_int32, _i, _i, _i, _i]),
    ("opusgpu_silk_nsq_workspace_bytes", C.c_size_t, [_i]),
    ("opusgpu_silk_nsq_batch", _i, [_vp, _vp, _vp, _i, _vp, C.c_size_t, _vp]),
    ("opusgpu_silk_nsq_del_dec_workspace_bytes", C.c_size_t, [_i]),
    ("opusgpu_silk_nsq_del_dec_batch", _i, [_vp, _vp, _vp, _i, _vp, C.c_size_t, _vp]),
    ("opusgpu_silk_bad_records", _i, [_vp]),
    ("opusgpu_silk_find_lpc_batch", _i, [_vp, _vp, _i, _vp]),
    ("opusgpu_silk_find_LPC_FIX", None, [_vp, _vp, _vp, C.c_int32]),
    ("opusgpu_silk_process_nlsfs_batch", _i, [_vp, _vp, _i, _vp]),
    ("opusgpu_silk_process_NLSFs", None, [_vp, _vp, _vp, _vp]),
    ("opusgpu_silk_residual_energy_batch", _i, [_vp, _vp, _i, _vp]),
    ("opusgpu_silk_find_pred_coefs_batch", _i, [_vp, _vp, _i, _vp]),
    ("opusgpu_silk_find_pred_coefs_FIX", None, [_vp, _vp, _vp, _vp, _i]),
    ("opusgpu_silk_process_gains_batch", _i, [_vp, _vp, _i, _vp]),
    ("opusgpu_silk_noise_shape_analysis_batch", _i, [_vp, _vp, _i, _vp]),
    ("opusgpu_silk_prefilter_batch", _i, [_vp, _vp, _vp, _i, _vp]),
    ("opusgpu_silk_find_pitch_lags_batch", _i, [_vp, _vp, _i, _vp]),
    ("opusgpu_silk_find_pitch_lags_FIX", None, [_vp, _vp, _vp, _vp, _i]),
    ("opusgpu_silk_encode_bits_batch", _i, [_vp, _vp, _vp, _i, _vp]),
    ("opusgpu_silk_vad_batch", _i, [_vp, _vp, _vp, _i, _vp]),
    ("opusgpu_silk_rate_control_batch", _i, [_vp, _vp, _i, _vp]),
    ("opusgpu_silk_encode_frames_batch", _i, [_vp, _i, _i, _i, _i, _vp]),
    ("opusgpu_silk_stream_carry_in", _i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    ("opusgpu_silk_stream_carry_out", _i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    ("opusgpu_silk_encode_frames_cbr_workspace_bytes", C.c_size_t, [_i]),
    ("opusgpu_silk_encode_frames_cbr_batch", _i, [_vp, _vp, _i, _i, _i, _i, _vp, C.c_size_t, _vp, _vp]),
    ("opusgpu_silk_VAD_GetSA_Q8_c", _i, [_vp, _vp]),
    ("opusgpu_silk_noise_shape_analysis_FIX", None, [_vp, _vp, _vp, _vp, _i]),
    ("opusgpu_silk_process_gains_FIX", None, [_vp, _vp, _i]),
    ("opusgpu_silk_prefilter_FIX", None, [_vp, _vp, _vp, _vp]),
    ("opusgpu_silk_encode_indices", None, [_vp, _vp, _i, _i, _i]),
    ("opusgpu_silk_encode_pulses", None, [_vp, _i, _i, _vp, _i]),
    ("opusgpu_silk_encode_frame_FIX", _i, [_vp, _vp, _vp, _i, _i, _i]),
    ("opusgpu_silk_residual_energy_FIX", None, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i]),
    ("opusgpu_celt_dec_state_size", _i, []),
    ("opusgpu_celt_dec_state_init", _i, [_vp, _i, _vp]),
    ("opusgpu_decode_batch", _i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    ("opusgpu_encoder_create", _vp, [_i, _i, _i, _vp]),
    ("opusgpu_encoder_ctl", _i, None),          # variadic
    ("opusgpu_encode", _i, [_vp, _vp, _i, _vp, _i]),
    ("opusgpu_encoder_destroy", None, [_vp]),
    ("opusgpu_decoder_create", _vp, [_i, _i, _vp]),
    ("opusgpu_decode", _i, [_vp, _vp, _i, _vp, _i, _i]),
    ("opusgpu_decoder_ctl", _i, None),          # variadic
    ("opusgpu_decoder_destroy", None, [_vp]),
    ("opusgpu_opus_ifft", None, [_vp, _vp, _vp]),
    ("opusgpu_comb_filter_const", None, [_vp, _vp, _i, _i, _i, _i, _i]),
    ("opusgpu_exp_rotation1", None, [_vp, _i, _i, _i, _i]),
    ("opusgpu_renormalise_vector", None, [_vp, _i, _i, _i]),
    ("opusgpu_quant_all_bands", None, [_i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, C.c_int32, C.c_int32, _vp, _i, _i, _vp, _i]),
    ("opusgpu_ec_enc_script", _i, [_vp, _vp, _i]),
    ("opusgpu_ec_dec_script", _i, [_vp, _vp, _i, _vp]),
    ("opusgpu_silk_NSQ", None, [_vp] * 13 + [_i, _i]),
    ("opusgpu_silk_NSQ_del_dec", None, [_vp] * 13 + [_i, _i]),
]

# include/opusgpu_diag.h: the stage-stamp builds, in a library of their own (tools/stage_profile*.py only)
DIAG_LIB_PATH = os.path.join(_HERE, "libopusgpu_diag.so")
DIAG_SYMBOLS = [
    ("opusgpu_decode_lane_diag", _i, [_vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp]),
    ("opusgpu_back_lane_diag", _i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp]),
    ("opusgpu_encode_batch_diag", _i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _vp, C.c_size_t, _vp, _vp]),
]

_lib = None
_diag = None


class OpusGpuError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        super().__init__("%s: %s (%d)" % (where, strerror(code), code))


def load():
    """Return the loaded library, raising if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "concentus_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C concentus_amd/csrc` (no CPU fallback exists)" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            if args is not None:
                fn.argtypes = args
        _lib = lib
    return _lib


def load_diag():
    """The diagnostic library (stage stamps); raises if it has not been built (`make -C concentus_amd/csrc diag`)."""
    global _diag
    if _diag is None:
        load()
        if not os.path.exists(DIAG_LIB_PATH):
            raise ImportError("concentus_amd: %s is missing -- `make -C concentus_amd/csrc diag`" % DIAG_LIB_PATH)
        lib = C.CDLL(DIAG_LIB_PATH)
        for name, res, args in DIAG_SYMBOLS:
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _diag = lib
    return _diag


def strerror(code):
    return load().opusgpu_strerror(code).decode()


def check(code, where):
    if code < 0:
        raise OpusGpuError(code, where)
    return code


def current_stream_handle():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)

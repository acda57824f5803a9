"""ctypes access to the host-emulation build of the frame kernel sources (tests/emu) -- TEST INFRASTRUCTURE."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "tests", "emu", "libcelt_emu.so")
LANE_PATH = os.path.join(ROOT, "tests", "emu", "libcelt_lane_emu.so")
HOST_CLANG = "/opt/rocm/lib/llvm/bin/clang++"         # the lane build's address-space qualifiers and ext vectors need clang
_lib = None
_lane = None


class Config(C.Structure):
    """opusgpu_celt_config (concentus_amd/csrc/celt_state.h)"""
    _fields_ = [(n, C.c_int32) for n in
                "channels bitrate vbr constrained_vbr complexity lsb_depth loss_rate max_data_bytes".split()]


class State(C.Structure):
    """opusgpu_celt_state (concentus_amd/csrc/celt_state.h)"""
    _fields_ = [
        ("hp_mem", C.c_int32 * 4), ("rng", C.c_uint32), ("spread_decision", C.c_int32), ("delayedIntra", C.c_int32),
        ("tonal_average", C.c_int32), ("lastCodedBands", C.c_int32), ("hf_average", C.c_int32),
        ("tapset_decision", C.c_int32), ("prefilter_period", C.c_int32), ("prefilter_gain", C.c_int32),
        ("prefilter_tapset", C.c_int32), ("consec_transient", C.c_int32), ("preemph_memE", C.c_int32 * 2),
        ("vbr_reservoir", C.c_int32), ("vbr_drift", C.c_int32), ("vbr_offset", C.c_int32), ("vbr_count", C.c_int32),
        ("overlap_max", C.c_int32), ("stereo_saving", C.c_int32), ("intensity", C.c_int32), ("spec_avg", C.c_int32),
        ("reserved", C.c_int32 * 5),
        ("oldBandE", C.c_int16 * 42), ("oldLogE", C.c_int16 * 42), ("oldLogE2", C.c_int16 * 42), ("pad16", C.c_int16 * 2),
        ("in_mem", C.c_int32 * 240), ("prefilter_mem", C.c_int32 * 2048),
    ]


def fresh_states(n):
    """n stream states as after opus_encoder_create (OPUS_RESET_STATE, celt_encoder.c:2443-2462)."""
    arr = np.zeros((n, C.sizeof(State)), np.uint8)
    for s in range(n):
        st = State.from_buffer(arr[s])
        st.spread_decision = 2
        st.delayedIntra = 1
        st.tonal_average = 256
        for k in range(42):
            st.oldLogE[k] = -28672
            st.oldLogE2[k] = -28672
    return arr


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(ROOT, "tests", "emu", "celt_emu.cpp")
        hdrs = [os.path.join(ROOT, "concentus_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "concentus_amd", "csrc")) if f.endswith(".h")]
        newest = max(os.path.getmtime(f) for f in hdrs + [src])
        if not os.path.exists(PATH) or os.path.getmtime(PATH) < newest:
            if any(k.startswith(("ROCP", "ROCPROFILER", "HSA_TOOLS")) for k in os.environ):
                raise RuntimeError("%s must be built before the profiler starts (a plain python3 run does it)" % PATH)
            subprocess.check_call(["g++", "-O2", "-fwrapv", "-std=c++17", "-shared", "-fPIC", "-w", "-o", PATH, src])
        _lib = C.CDLL(PATH)
        assert _lib.emu_sizeof_state() == C.sizeof(State)
    return _lib


def lane_lib():
    """Host build of the LANE-PER-FRAME variant of the sources (tests/emu/celt_lane_emu.cpp): the code paths of
    celt_back_lane_kernel / celt_decode_lane_kernel on a CPU, one frame at a time in one column of a 64-column LDS image."""
    global _lane
    if _lane is None:
        src = os.path.join(ROOT, "tests", "emu", "celt_lane_emu.cpp")
        d = os.path.join(ROOT, "concentus_amd", "csrc")
        newest = max(os.path.getmtime(f) for f in [os.path.join(d, f) for f in os.listdir(d) if f.endswith(".h")] + [src])
        if not os.path.exists(LANE_PATH) or os.path.getmtime(LANE_PATH) < newest:
            if any(k.startswith(("ROCP", "ROCPROFILER", "HSA_TOOLS")) for k in os.environ):
                raise RuntimeError("%s must be built before the profiler starts" % LANE_PATH)
            subprocess.check_call([HOST_CLANG, "-O1", "-fwrapv", "-std=c++17", "-shared", "-fPIC", "-w", "-o", LANE_PATH, src])
        _lane = C.CDLL(LANE_PATH)
    return _lane

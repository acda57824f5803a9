"""ctypes access to oracle/liboracle.so (our CPU restatement) -- TEST INFRASTRUCTURE."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "oracle", "liboracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(PATH):
            if any(k.startswith(("ROCP", "ROCPROFILER", "HSA_TOOLS")) for k in os.environ):
                raise RuntimeError("%s must be built before the profiler starts (a plain python3 run does it)" % PATH)
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"])
        _lib = C.CDLL(PATH)
    return _lib


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)

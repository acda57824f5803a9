"""CPU tier: the C-ABI library loads and exports every symbol include/opusgpu.h declares
(no compute calls -- there is no GPU here)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    names = set()
    for h in ("opusgpu.h", "opusgpu_silk.h", "opusgpu_hooks.h"):
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names |= set(re.findall(r"\b(opusgpu_\w+)\s*\(", src))
    return sorted(names)


def _ensure_built():
    from concentus_amd import lib
    if not os.path.exists(lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return lib


def test_header_symbols_all_exported():
    lib = _ensure_built()
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib.LIB_PATH], text=True)
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    missing = [s for s in _declared() if s not in exported]
    assert not missing, missing


def test_diag_library_is_separate():
    """The stage-stamp diagnostic builds live in libopusgpu_diag.so (include/opusgpu_diag.h), not in the product library."""
    lib = _ensure_built()
    src = open(os.path.join(ROOT, "include", "opusgpu_diag.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    declared = set(re.findall(r"\b(opusgpu_\w+)\s*\(", src))
    assert declared == {name for name, _, _ in lib.DIAG_SYMBOLS}
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib.LIB_PATH], text=True)
    assert not [l for l in out.splitlines() if "_diag" in l], "diagnostic code in the product library"
    if os.path.exists(lib.DIAG_LIB_PATH):
        out = subprocess.check_output(["nm", "-D", "--defined-only", lib.DIAG_LIB_PATH], text=True)
        exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
        assert declared <= exported


def test_python_binding_covers_header():
    lib = _ensure_built()
    bound = {name for name, _, _ in lib.SYMBOLS}
    assert set(_declared()) == bound
    lib.load()  # resolves every symbol via ctypes; raises AttributeError if one is absent


def test_strerror_and_version_without_gpu():
    lib = _ensure_built()
    L = lib.load()
    assert b"gfx950" in L.opusgpu_get_version_string()
    assert lib.strerror(0) == "success"
    assert lib.strerror(-1) == "invalid argument"
    assert lib.strerror(-99) == "unknown error"


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from concentus_amd import lib
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError):
        lib.load()


def test_compat_library_exports_the_libopus_names():
    """concentus_amd/compat/libopus.so: the link target for unedited callers of <opus.h> (src/opus_demo.c, the P/Invoke
    harness CSharp/ParityTest/TestDriver.cs:22-41) must export exactly the names they bind."""
    _ensure_built()
    path = os.path.join(ROOT, "concentus_amd", "compat", "libopus.so")
    assert os.path.exists(path), "make -C concentus_amd/csrc builds it"
    out = subprocess.check_output(["nm", "-D", "--defined-only", path], text=True)
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    need = {"opus_encoder_create", "opus_encoder_ctl", "opus_encode", "opus_encoder_destroy", "opus_decoder_create",
            "opus_decoder_ctl", "opus_decode", "opus_decoder_destroy", "opus_strerror", "opus_get_version_string",
            "opus_packet_get_samples_per_frame", "opus_packet_get_nb_frames", "opus_packet_get_nb_samples"}
    assert need <= exported, sorted(need - exported)
    import ctypes
    L = ctypes.CDLL(path)                                     # pulls libopusgpu.so in through its rpath
    L.opus_get_version_string.restype = ctypes.c_char_p
    assert b"gfx950" in L.opus_get_version_string()
    toc = (ctypes.c_ubyte * 2)(0xFC, 0)
    assert L.opus_packet_get_samples_per_frame(toc, 48000) == 960 and L.opus_packet_get_nb_frames(toc, 2) == 1


def test_chain_entry_point_rejects_bad_arguments_before_touching_a_gpu():
    """opusgpu_silk_encode_frames_batch checks its buffer table and geometry on the host first: NULL table, a missing stage buffer,
    12 kHz / three subframes -> OPUSGPU_BAD_ARG; zero frames -> OK. None of these reaches a HIP call, so they run here."""
    import ctypes as C
    lib = _ensure_built()
    from concentus_amd.silk_chain import ChainBufs
    L = C.CDLL(lib.LIB_PATH)
    f = L.opusgpu_silk_encode_frames_batch
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    assert f(None, 16, 4, 1, 8, None) == -1
    empty = ChainBufs()
    assert f(C.byref(empty), 16, 4, 1, 0, None) == 0
    assert f(C.byref(empty), 16, 4, 1, 8, None) == -1
    full = ChainBufs(*([C.c_void_p(4096)] * 18), C.c_size_t(0))               # never dereferenced: the geometry is rejected first
    assert f(C.byref(full), 12, 4, 1, 8, None) == -1
    assert f(C.byref(full), 16, 3, 0, 8, None) == -1
    assert f(C.byref(full), 16, 4, 0, -1, None) == -1

"""CPU tier: the LDS bank swizzle of the FFT scratch (concentus_amd/csrc/mdct_dev.h fsw<0>). (1) It is a bijection and the cheap
member addressing of the butterfly stages / the pre-swizzled bit-reversal table equal it for every point (tests/emu/
fft_swizzle_check.cpp, the device header compiled for the host). (2) Under the bank model of MI355X_MICROARCH.md
(tools/fft_swizzle_search.py) the stages of the long transform have at most 0.3 conflict cycles per useful LDS cycle -- the round-3
criterion -- where the plain layout has 2."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_member_addressing_equals_the_swizzle_everywhere():
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "chk")
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-w", "-o", exe, os.path.join(ROOT, "tests", "emu", "fft_swizzle_check.cpp")])
        out = subprocess.run([exe], capture_output=True, text=True)
        assert out.returncode == 0 and "mismatches 0" in out.stdout, out.stdout + out.stderr


def test_bank_model_conflict_ratio():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fft_swizzle_search as fs
    t, c = fs.total(fs.stages(True), fs.fsw)
    assert c / (t - c) < 0.3, (t, c)
    t0, c0 = fs.total(fs.stages(False), lambda e: e)
    assert c0 / (t0 - c0) > 1.5 and t < t0 / 3, (t0, c0, t)

"""Distinct SILK function-boundary records at BASELINE config #4's size -- TEST / BENCH INPUT GENERATION.

SURVEY.md 8d defines config #4's input as "65 536 function-boundary records captured from the oracle encoding 16 kHz
mono voice at 32 kb/s, VOIP". The 80 committed golden records cover parity on real captures; this module produces the
full-size corpus the same way -- the UNMODIFIED reference encoder (oracle/_ref/libopus_ref_silkcap.so, built by
oracle/Makefile with --wrap capture shims, oracle/ref_silk_capture.c) encodes synthetic speech-like audio and every call
of silk_burg_modified_c / silk_NSQ_c / silk_NSQ_del_dec_c is recorded with its arguments AND its results, so the
corpus carries the reference's own outputs for every record. Nothing is tiled: every record comes from a different
frame of a different piece of audio (voiced / unvoiced / pauses, pitch 70-320 Hz, changing formants and levels).

The capture library keeps its buffers in globals, so segments are captured in worker PROCESSES; the records land in
memory-mapped .npy files under a cache directory (default $CONCENTUS_SILK_CACHE or /tmp/concentus_silk_corpus), which a
later call re-opens instead of re-encoding. Nothing here is imported by the product package.
"""
import ctypes as C
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CAPLIB = os.path.join(ROOT, "oracle", "_ref", "libopus_ref_silkcap.so")
FS = 16000
FRAME = 320
# (input rate, samples per opus_encode call); "wb40": two SILK frames per packet, the second one coded conditionally
VARIANTS = {"wb20": (16000, 320), "nb20": (8000, 160), "wb10": (16000, 160), "wb40": (16000, 640), "wb20cbr": (16000, 320),
            "nb20cbr": (8000, 160), "wb20lo": (16000, 320)}
# encoder settings a variant changes: "...cbr" = OPUS_SET_VBR(0) (silk_encode_frame_FIX runs its bitrate loop on every frame);
# "wb20lo" = VBR with max_data_bytes 50 (the loop runs where the first pass overshoots maxBits)
VARIANT_CTLS = {"wb20cbr": ((4006, 0),), "nb20cbr": ((4006, 0), (4002, 16000))}
VARIANT_MAX_BYTES = {"wb20lo": 50}
MAX_PASSES = 7              # silk_encode_frame_FIX: iter 0 .. maxIter = 6
SEG_FRAMES = 2048           # frames per captured segment (one fresh encoder each)
WARMUP = 8                  # leading frames of a segment whose records are dropped (encoder start-up)

SIZES = {"burg_in": 784, "burg_out": 72, "nsq_in": 1640, "nsq_state": 4380, "nsq_out": 320, "dd_in": 1648, "dd_out": 324,
         "lpc_in": 832, "lpc_out": 40, "nlsf_in": 96, "nlsf_out": 120, "resnrg_in": 864, "resnrg_out": 40,
         "fpc_in": 2688, "fpc_out": 208, "gains_in": 112, "gains_out": 56,
         "shape_in": 1696, "shape_out": 384, "prefilter_in": 896, "prefilter_state": 1116, "prefilter_out": 1296,
         "pitch_in": 1408, "pitch_out": 1392, "bits_in": 416, "ec_state": 1328, "bits_out": 16,
         "vad_in": 656, "vad_state": 112, "vad_out": 32,
         "frame_args": 16, "frame_misc": 356, "frame_passes": 256}


def available():
    return os.path.exists(CAPLIB)


def synth_voice(nsamples, seed):
    """Speech-like mono int16 at 16 kHz: glottal pulse train with a wandering pitch (70-320 Hz) or noise excitation
    through three formant resonators whose centre frequencies change every 120-400 ms; utterances of 0.4-2.5 s at
    random levels separated by pauses."""
    from scipy.signal import lfilter
    rng = np.random.default_rng(seed)
    n = nsamples
    # pitch contour: random knots every 100 ms, linearly interpolated
    knots = np.clip(np.cumsum(rng.normal(0, 12, n // 1600 + 2)) + rng.uniform(90, 240), 70, 320)
    f0 = np.interp(np.arange(n), np.arange(len(knots)) * 1600, knots)
    phase = np.cumsum(f0 / FS)
    pulses = (np.diff(np.floor(phase), prepend=0) > 0).astype(np.float64)
    # segmentation into voiced / unvoiced / pause with a level per utterance
    voiced = np.zeros(n)
    gain = np.zeros(n)
    pos = 0
    while pos < n:
        utt = int(rng.uniform(0.4, 2.5) * FS)
        level = 10 ** (rng.uniform(-26, 0) / 20)
        end = min(n, pos + utt)
        q = pos
        while q < end:
            v = rng.random() < 0.65
            d = int(rng.uniform(0.08, 0.5) * FS) if v else int(rng.uniform(0.03, 0.18) * FS)
            voiced[q:min(end, q + d)] = 1.0 if v else 0.0
            gain[q:min(end, q + d)] = level * (1.0 if v else rng.uniform(0.15, 0.6))
            q += d
        pos = end + int(max(0.0, rng.normal(0.15, 0.2)) * FS)
    exc = pulses * voiced * 1.0 + rng.normal(0, 0.05, n) * (1 - voiced) + rng.normal(0, 0.002, n)
    exc *= gain
    # time-varying formants: constant within a chunk, filter memories carried across chunks
    y = np.empty(n)
    zi = [np.zeros(2), np.zeros(2), np.zeros(2)]
    vowels = np.array([[730, 1090, 2440], [270, 2290, 3010], [300, 870, 2240], [530, 1840, 2480], [660, 1720, 2410],
                       [440, 1020, 2240], [490, 1350, 1690], [400, 2000, 2550]], dtype=np.float64)
    pos = 0
    while pos < n:
        d = int(rng.uniform(0.12, 0.4) * FS)
        seg = exc[pos:pos + d]
        fm = vowels[rng.integers(len(vowels))] * rng.uniform(0.9, 1.12)
        bw = np.array([rng.uniform(60, 140), rng.uniform(60, 120), rng.uniform(110, 200)])
        for k in range(3):
            r = np.exp(-np.pi * bw[k] / FS)
            a = [1.0, -2 * r * np.cos(2 * np.pi * min(fm[k], 7000) / FS), r * r]
            seg, zi[k] = lfilter([1 - r], a, seg, zi=zi[k])
        y[pos:pos + len(seg)] = seg
        pos += d
    peak = np.abs(y).max() + 1e-12
    return np.clip(y / peak * 24000, -32768, 32767).astype(np.int16)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _capture_segment(args):
    """Worker: encode SEG_FRAMES + WARMUP frames of synth_voice(seed) with a fresh reference encoder and write the
    records of frames [WARMUP, WARMUP + take) into rows [row0, row0 + take) of the corpus files."""
    cache, kind, seed, complexity, row0, take, total, variant = args
    fs, frame = VARIANTS[variant]
    lib = C.CDLL(CAPLIB)
    lib.opus_encoder_create.restype = C.c_void_p
    lib.opus_encoder_ctl.argtypes = [C.c_void_p, C.c_int, C.c_int]
    nfr = take + WARMUP
    pcm = synth_voice(nfr * frame * (FS // fs), seed)[::FS // fs]          # 8 kHz: every second sample of the 16 kHz synthesis
    cap = 3 * nfr + 16
    if kind.startswith("chain"):
        # up to MAX_PASSES quantiser / coder calls per frame where the bitrate loop iterates
        lib.refcap_start_chain((8 if variant in VARIANT_CTLS or variant in VARIANT_MAX_BYTES else 4) * nfr + 16)
    elif kind == "dd":
        lib.refcap_start_dd(cap)
    elif kind == "lpc":
        lib.refcap_start_lpc(cap)
    elif kind == "pred":
        lib.refcap_start_pred(cap)
    elif kind == "fpc":
        lib.refcap_start_fpc(cap)
    elif kind == "gains":
        lib.refcap_start_gains(cap)
    elif kind == "shape":
        lib.refcap_start_shape(cap)
    elif kind == "prefilter":
        lib.refcap_start_prefilter(cap)
    elif kind == "pitch":
        lib.refcap_start_pitch(cap)
    elif kind == "bits":
        lib.refcap_start_bits(cap)
    elif kind == "vad":
        lib.refcap_start_vad(cap)
    else:
        lib.refcap_start(cap)
    err = C.c_int()
    enc = C.c_void_p(lib.opus_encoder_create(fs, 1, 2048, C.byref(err)))          # OPUS_APPLICATION_VOIP
    assert enc and err.value == 0
    for req, v in ((4002, 32000), (4006, 1), (4020, 0), (4010, complexity), (4012, 0), (4016, 0), (4014, 0), (4036, 16)) + VARIANT_CTLS.get(variant, ()):
        lib.opus_encoder_ctl(enc, req, v)
    out = (C.c_ubyte * 1500)()
    for f in range(nfr):
        fr = np.ascontiguousarray(pcm[f * frame:(f + 1) * frame])
        assert lib.opus_encode(enc, _p(fr), frame, out, VARIANT_MAX_BYTES.get(variant, 1500)) > 0
    files = _files(cache, kind, total, mode="r+")
    if kind.startswith("chain"):
        dd = kind == "chain_dd"

        def fids(k, n):
            a = np.zeros(max(n, 1), np.int32)
            lib.refcap_get_frame_ids(k, _p(a), n)
            return a[:n]

        def grab(n, getter, sizes):
            bufs = [np.zeros((n, SIZES[z]), np.uint8) for z in sizes]
            getter(*[_p(b) for b in bufs])
            return bufs
        got = {}
        n = lib.refcap_count_pitch(); got["pitch"] = (grab(n, lib.refcap_get_pitch, ("pitch_in", "pitch_out")), fids(0, n))
        n = lib.refcap_count_shape(); got["shape"] = (grab(n, lib.refcap_get_shape, ("shape_in", "shape_out")), fids(1, n))
        n = lib.refcap_count_fpc(); got["fpc"] = (grab(n, lib.refcap_get_fpc, ("fpc_in", "fpc_out")), fids(2, n))
        n = lib.refcap_count_gains(); got["gains"] = (grab(n, lib.refcap_get_gains, ("gains_in", "gains_out")), fids(3, n))
        n = lib.refcap_count_prefilter()
        got["prefilter"] = (grab(n, lib.refcap_get_prefilter, ("prefilter_in", "prefilter_state", "prefilter_state", "prefilter_out")), fids(4, n))
        if dd:
            n = lib.refcap_count_dd(); got["q"] = (grab(n, lib.refcap_get_dd, ("dd_in", "nsq_state", "nsq_state", "dd_out")), fids(6, n))
        else:
            nb, n = lib.refcap_count_burg(), lib.refcap_count_nsq()
            b = [np.zeros((nb, SIZES["burg_in"]), np.uint8), np.zeros((nb, SIZES["burg_out"]), np.uint8)]
            q = [np.zeros((n, SIZES[z]), np.uint8) for z in ("nsq_in", "nsq_state", "nsq_state", "nsq_out")]
            lib.refcap_get(*[_p(a) for a in b + q])
            got["q"] = (q, fids(5, n))
        # the frame's two entropy-coding calls as ONE record: the indices record with the pulses of the pulses record, which = 3;
        # coder before silk_encode_indices -> coder after silk_encode_pulses
        ni, npl = lib.refcap_count_bits(0), lib.refcap_count_bits(1)
        bi = [np.zeros((ni, SIZES[z]), np.uint8) for z in ("bits_in", "ec_state", "ec_state", "bits_out")]
        bp = [np.zeros((npl, SIZES[z]), np.uint8) for z in ("bits_in", "ec_state", "ec_state", "bits_out")]
        lib.refcap_get_bits(0, *[_p(b) for b in bi])
        lib.refcap_get_bits(1, *[_p(b) for b in bp])
        got["bits_idx"] = (bi, fids(7, ni))
        got["bits_pls"] = (bp, fids(8, npl))
        first = {}
        for key, (_, fid) in got.items():
            # index of the first record of every frame number (frame numbers start at 1)
            order = np.argsort(fid, kind="stable")
            uniq, start = np.unique(fid[order], return_index=True)
            first[key] = dict(zip(uniq.tolist(), order[start].tolist()))
        frames = [f for f in range(WARMUP + 1, nfr + 1) if all(f in first[k] for k in first)][:take]
        assert len(frames) == take, (len(frames), take)
        names = {"pitch": ("c_pitch_in", "c_pitch_out"), "shape": ("c_shape_in", "c_shape_out"), "fpc": ("c_fpc_in", "c_fpc_out"),
                 "gains": ("c_gains_in", "c_gains_out"),
                 "prefilter": ("c_prefilter_in", "c_prefilter_state_in", "c_prefilter_state_out", "c_prefilter_out"),
                 "q": ("c_q_in", "c_q_state_in", "c_q_state_out", "c_q_out")}
        for key, (bufs, _) in got.items():
            if key.startswith("bits"):
                continue
            sel = np.array([first[key][f] for f in frames])
            for name, b in zip(names[key], bufs):
                files[name][row0:row0 + take] = b[sel]
        si = np.array([first["bits_idx"][f] for f in frames])
        sp = np.array([first["bits_pls"][f] for f in frames])
        both = bi[0][si].copy()
        both[:, :320] = bp[0][sp][:, :320]
        both[:, 348 + 60:348 + 64].view(np.int32)[:, 0] = 3
        assert np.array_equal(bi[2][si], bp[1][sp]), "indices-out must be pulses-in on the same coder"
        files["c_bits_in"][row0:row0 + take] = both
        files["c_ec_in"][row0:row0 + take] = bi[1][si]
        files["c_ec_out"][row0:row0 + take] = bp[2][sp]
        files["c_bits_out"][row0:row0 + take] = bi[3][si]
        # silk_encode_frame_FIX as a whole: arguments, what it leaves behind, and every quantise + code pass of its bitrate loop
        # (gain indices coded, gains and Lambda_Q10 quantised with, ec_tell() after the pass)
        if ROOT not in sys.path:
            sys.path.insert(0, ROOT)
        from concentus_amd import silk as S
        nfrm = lib.refcap_count_frame()
        assert [lib.refcap_sizes_frame(k) for k in range(4)] == [SIZES["frame_args"], SIZES["ec_state"], SIZES["nsq_state"], SIZES["frame_misc"]]
        fr_bufs = grab(nfrm, lib.refcap_get_frame, ("frame_args", "ec_state", "nsq_state", "frame_misc"))
        ffid = fids(9, nfrm)
        fsel = {int(f): k for k, f in enumerate(ffid)}
        sel = np.array([fsel[f] for f in frames])
        for name, b in zip(("c_frame_args", "c_frame_ec", "c_frame_nsq", "c_frame_misc"), fr_bufs):
            files[name][row0:row0 + take] = b[sel]
        qb, qfid = got["q"][0][0], got["q"][1]
        pfid = got["bits_pls"][1]
        ec_after = bp[2]
        tell = ec_after[:, 16:20].copy().view(np.int32)[:, 0] - np.array([int(v).bit_length() for v in ec_after[:, 24:28].copy().view(np.uint32)[:, 0]])
        g_off, l_off = S.NsqIn.Gains_Q16.offset, S.NsqIn.Lambda_Q10.offset
        passes = np.zeros((take, SIZES["frame_passes"]), np.uint8)
        pv = passes.view(np.int32)
        for row, f in enumerate(frames):
            qi = np.flatnonzero(qfid == f)
            pi = np.flatnonzero(pfid == f)
            ii = np.flatnonzero(got["bits_idx"][1] == f)
            assert len(qi) == len(pi) == len(ii) == fr_bufs[0][fsel[f]].view(np.int32)[3] <= MAX_PASSES, (f, len(qi), len(pi))
            pv[row, 0] = len(qi)
            for k in range(len(qi)):
                base = 1 + 7 * k
                passes[row, 4 * base:4 * base + 4] = bi[0][ii[k]][320:324]                       # GainsIndices coded in pass k
                pv[row, base + 1] = tell[pi[k]]
                pv[row, base + 2:base + 6] = qb[qi[k]][g_off:g_off + 16].view(np.int32)
                pv[row, base + 6] = qb[qi[k]][l_off:l_off + 4].view(np.int32)[0]
        files["c_frame_passes"][row0:row0 + take] = passes
        for f in files.values():
            f.flush()
        return take
    if kind == "vad":
        nv = lib.refcap_count_vad()
        assert nv >= nfr, (nv, nfr)
        assert [lib.refcap_sizes_vad(k) for k in range(4)] == [SIZES["vad_in"], SIZES["vad_state"], SIZES["vad_out"], SIZES["vad_state"]]
        bufs = [np.zeros((nv, SIZES["vad_in"]), np.uint8), np.zeros((nv, SIZES["vad_state"]), np.uint8),
                np.zeros((nv, SIZES["vad_state"]), np.uint8), np.zeros((nv, SIZES["vad_out"]), np.uint8)]
        lib.refcap_get_vad(*[_p(b) for b in bufs])
        for name, b in zip(("vad_in", "vad_state_in", "vad_state_out", "vad_out"), bufs):
            files[name][row0:row0 + take] = b[:take]          # from the first frame: the start-up of the noise-level tracker counts
    elif kind == "bits":
        assert [lib.refcap_sizes_bits(k) for k in range(3)] == [SIZES["bits_in"], SIZES["ec_state"], SIZES["bits_out"]]
        for which, tag in ((0, "idx"), (1, "pls")):
            nb_ = lib.refcap_count_bits(which)
            assert nb_ >= nfr, (which, nb_, nfr)
            bufs = [np.zeros((nb_, SIZES["bits_in"]), np.uint8), np.zeros((nb_, SIZES["ec_state"]), np.uint8),
                    np.zeros((nb_, SIZES["ec_state"]), np.uint8), np.zeros((nb_, SIZES["bits_out"]), np.uint8)]
            lib.refcap_get_bits(which, *[_p(b) for b in bufs])
            for name, b in zip(("bits_%s_in" % tag, "bits_%s_ec_in" % tag, "bits_%s_ec_out" % tag, "bits_%s_out" % tag), bufs):
                files[name][row0:row0 + take] = b[WARMUP:WARMUP + take]
    elif kind == "pitch":
        nt = lib.refcap_count_pitch()
        assert nt >= nfr, (nt, nfr)
        assert (lib.refcap_sizes_pitch(0), lib.refcap_sizes_pitch(1)) == (SIZES["pitch_in"], SIZES["pitch_out"])
        bufs = [np.zeros((nt, SIZES["pitch_in"]), np.uint8), np.zeros((nt, SIZES["pitch_out"]), np.uint8)]
        lib.refcap_get_pitch(*[_p(b) for b in bufs])
        for name, b in zip(("pitch_in", "pitch_out"), bufs):
            files[name][row0:row0 + take] = b[WARMUP:WARMUP + take]
    elif kind == "prefilter":
        nx = lib.refcap_count_prefilter()
        assert nx >= nfr, (nx, nfr)
        assert [lib.refcap_sizes_prefilter(k) for k in range(4)] == [SIZES["prefilter_in"], SIZES["prefilter_state"], SIZES["prefilter_out"],
                                                                     SIZES["prefilter_state"]]
        bufs = [np.zeros((nx, SIZES["prefilter_in"]), np.uint8), np.zeros((nx, SIZES["prefilter_state"]), np.uint8),
                np.zeros((nx, SIZES["prefilter_state"]), np.uint8), np.zeros((nx, SIZES["prefilter_out"]), np.uint8)]
        lib.refcap_get_prefilter(*[_p(b) for b in bufs])
        for name, b in zip(("prefilter_in", "prefilter_state_in", "prefilter_state_out", "prefilter_out"), bufs):
            files[name][row0:row0 + take] = b[WARMUP:WARMUP + take]
    elif kind == "shape":
        nsh = lib.refcap_count_shape()
        assert nsh >= nfr, (nsh, nfr)
        assert (lib.refcap_sizes_shape(0), lib.refcap_sizes_shape(1)) == (SIZES["shape_in"], SIZES["shape_out"])
        bufs = [np.zeros((nsh, SIZES["shape_in"]), np.uint8), np.zeros((nsh, SIZES["shape_out"]), np.uint8)]
        lib.refcap_get_shape(*[_p(b) for b in bufs])
        for name, b in zip(("shape_in", "shape_out"), bufs):
            files[name][row0:row0 + take] = b[WARMUP:WARMUP + take]
    elif kind == "gains":
        ng = lib.refcap_count_gains()
        assert ng >= nfr, (ng, nfr)
        assert (lib.refcap_sizes_gains(0), lib.refcap_sizes_gains(1)) == (SIZES["gains_in"], SIZES["gains_out"])
        bufs = [np.zeros((ng, SIZES["gains_in"]), np.uint8), np.zeros((ng, SIZES["gains_out"]), np.uint8)]
        lib.refcap_get_gains(*[_p(b) for b in bufs])
        for name, b in zip(("gains_in", "gains_out"), bufs):
            files[name][row0:row0 + take] = b[WARMUP:WARMUP + take]
    elif kind == "fpc":
        nf = lib.refcap_count_fpc()
        assert nf >= nfr, (nf, nfr)
        assert (lib.refcap_sizes_fpc(0), lib.refcap_sizes_fpc(1)) == (SIZES["fpc_in"], SIZES["fpc_out"])
        bufs = [np.zeros((nf, SIZES["fpc_in"]), np.uint8), np.zeros((nf, SIZES["fpc_out"]), np.uint8)]
        lib.refcap_get_fpc(*[_p(b) for b in bufs])
        for name, b in zip(("fpc_in", "fpc_out"), bufs):
            files[name][row0:row0 + take] = b[WARMUP:WARMUP + take]
    elif kind == "pred":
        npr, ne = lib.refcap_count_pred(0), lib.refcap_count_pred(1)
        assert npr >= nfr and ne >= nfr, (npr, ne, nfr)
        bufs = [np.zeros((npr, SIZES["nlsf_in"]), np.uint8), np.zeros((npr, SIZES["nlsf_out"]), np.uint8),
                np.zeros((ne, SIZES["resnrg_in"]), np.uint8), np.zeros((ne, SIZES["resnrg_out"]), np.uint8)]
        lib.refcap_get_pred(*[_p(b) for b in bufs])
        for name, b in zip(("nlsf_in", "nlsf_out", "resnrg_in", "resnrg_out"), bufs):
            files[name][row0:row0 + take] = b[WARMUP:WARMUP + take]
    elif kind == "lpc":
        nl = lib.refcap_count_lpc()
        assert nl >= nfr, (nl, nfr)
        bufs = [np.zeros((nl, SIZES["lpc_in"]), np.uint8), np.zeros((nl, SIZES["lpc_out"]), np.uint8)]
        lib.refcap_get_lpc(*[_p(b) for b in bufs])
        for name, b in zip(("lpc_in", "lpc_out"), bufs):
            files[name][row0:row0 + take] = b[WARMUP:WARMUP + take]
    elif kind == "dd":
        nd = lib.refcap_count_dd()
        assert nd >= nfr, (nd, nfr)
        bufs = [np.zeros((nd, SIZES["dd_in"]), np.uint8), np.zeros((nd, SIZES["nsq_state"]), np.uint8),
                np.zeros((nd, SIZES["nsq_state"]), np.uint8), np.zeros((nd, SIZES["dd_out"]), np.uint8)]
        lib.refcap_get_dd(*[_p(b) for b in bufs])
        for name, b in zip(("dd_in", "dd_state_in", "dd_state_out", "dd_out"), bufs):
            files[name][row0:row0 + take] = b[WARMUP:WARMUP + take]
    else:
        nb, nn = lib.refcap_count_burg(), lib.refcap_count_nsq()
        assert nn >= nfr and nb >= nfr, (nb, nn, nfr)
        bufs = [np.zeros((nb, SIZES["burg_in"]), np.uint8), np.zeros((nb, SIZES["burg_out"]), np.uint8),
                np.zeros((nn, SIZES["nsq_in"]), np.uint8), np.zeros((nn, SIZES["nsq_state"]), np.uint8),
                np.zeros((nn, SIZES["nsq_state"]), np.uint8), np.zeros((nn, SIZES["nsq_out"]), np.uint8)]
        lib.refcap_get(*[_p(b) for b in bufs])
        skip_b = nb - (nn - WARMUP) if nb > nn else WARMUP       # burg may run twice per frame: keep the trailing ones
        skip_b = max(WARMUP, min(skip_b, nb - take))
        for name, b in zip(("burg_in", "burg_out"), bufs[:2]):
            files[name][row0:row0 + take] = b[skip_b:skip_b + take]
        for name, b in zip(("nsq_in", "nsq_state_in", "nsq_state_out", "nsq_out"), bufs[2:]):
            files[name][row0:row0 + take] = b[WARMUP:WARMUP + take]
    for f in files.values():
        f.flush()
    return take


_CHAIN_LAYOUT = (("c_pitch_in", "pitch_in"), ("c_pitch_out", "pitch_out"), ("c_shape_in", "shape_in"), ("c_shape_out", "shape_out"),
                 ("c_fpc_in", "fpc_in"), ("c_fpc_out", "fpc_out"), ("c_gains_in", "gains_in"), ("c_gains_out", "gains_out"),
                 ("c_prefilter_in", "prefilter_in"), ("c_prefilter_state_in", "prefilter_state"), ("c_prefilter_state_out", "prefilter_state"),
                 ("c_prefilter_out", "prefilter_out"), ("c_q_state_in", "nsq_state"), ("c_q_state_out", "nsq_state"),
                 ("c_bits_in", "bits_in"), ("c_ec_in", "ec_state"), ("c_ec_out", "ec_state"), ("c_bits_out", "bits_out"),
                 ("c_frame_args", "frame_args"), ("c_frame_ec", "ec_state"), ("c_frame_nsq", "nsq_state"), ("c_frame_misc", "frame_misc"),
                 ("c_frame_passes", "frame_passes"))
_LAYOUT = {
    "nsq": (("burg_in", "burg_in"), ("burg_out", "burg_out"), ("nsq_in", "nsq_in"), ("nsq_state_in", "nsq_state"),
            ("nsq_state_out", "nsq_state"), ("nsq_out", "nsq_out")),
    "dd": (("dd_in", "dd_in"), ("dd_state_in", "nsq_state"), ("dd_state_out", "nsq_state"), ("dd_out", "dd_out")),
    "lpc": (("lpc_in", "lpc_in"), ("lpc_out", "lpc_out")),
    "fpc": (("fpc_in", "fpc_in"), ("fpc_out", "fpc_out")),
    "gains": (("gains_in", "gains_in"), ("gains_out", "gains_out")),
    "shape": (("shape_in", "shape_in"), ("shape_out", "shape_out")),
    "pitch": (("pitch_in", "pitch_in"), ("pitch_out", "pitch_out")),
    "vad": (("vad_in", "vad_in"), ("vad_state_in", "vad_state"), ("vad_state_out", "vad_state"), ("vad_out", "vad_out")),
    "bits": (("bits_idx_in", "bits_in"), ("bits_idx_ec_in", "ec_state"), ("bits_idx_ec_out", "ec_state"), ("bits_idx_out", "bits_out"),
             ("bits_pls_in", "bits_in"), ("bits_pls_ec_in", "ec_state"), ("bits_pls_ec_out", "ec_state"), ("bits_pls_out", "bits_out")),
    # aligned capture of ONE encoder run: row r of every array belongs to the same frame (first call of the frame)
    "chain_nsq": _CHAIN_LAYOUT + (("c_q_in", "nsq_in"), ("c_q_out", "nsq_out")),
    "chain_dd": _CHAIN_LAYOUT + (("c_q_in", "dd_in"), ("c_q_out", "dd_out")),
    "prefilter": (("prefilter_in", "prefilter_in"), ("prefilter_state_in", "prefilter_state"), ("prefilter_state_out", "prefilter_state"),
                  ("prefilter_out", "prefilter_out")),
    "pred": (("nlsf_in", "nlsf_in"), ("nlsf_out", "nlsf_out"), ("resnrg_in", "resnrg_in"), ("resnrg_out", "resnrg_out")),
}


_TOOL_ENV = ("ROCP", "ROCPROFILER", "HSA_TOOLS", "ROCTRACER", "OMNITRACE")


def under_profiler():
    """True when this process was started by rocprofv3 (its tool libraries are preloaded into every child)."""
    return any(k.startswith(_TOOL_ENV) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")


def _files(cache, kind, n, mode):
    out = {}
    for name, sz in _LAYOUT[kind]:
        path = os.path.join(cache, "%s_%d.npy" % (name, n))
        if mode == "w+":
            out[name] = np.lib.format.open_memmap(path, mode="w+", dtype=np.uint8, shape=(n, SIZES[sz]))
        else:
            out[name] = np.load(path, mmap_mode=mode)
    return out


def corpus(n, kind="nsq", complexities=None, workers=None, cache=None, seed=20260401, variant="wb20", seg_frames=None):
    """n distinct records: kind "nsq" -> burg_in/burg_out/nsq_in/nsq_state_in/nsq_state_out/nsq_out captured at
    complexity 3 (silk_NSQ_c, control_codec.c:333-343); kind "dd" -> dd_in/dd_state_in/dd_state_out/dd_out captured at
    complexities 5 / 7 / 10 in turn (2 / 3 / 4 delayed-decision states). Returns read-only memory maps."""
    if not available():
        raise FileNotFoundError(CAPLIB)
    # kind "fpc": silk_find_pred_coefs_FIX whole (voiced and unvoiced frames), complexities as "lpc"
    # kind "pred": silk_process_NLSFs + silk_residual_energy_FIX (the tail of silk_find_pred_coefs_FIX), complexities as "lpc"
    # kind "lpc": silk_find_LPC_FIX at complexity 3 (no NLSF interpolation: Burg + A2NLSF) and 5 / 8 / 10 (interpolation search)
    # seg_frames: frames per segment = consecutive frames of ONE encoder (rows [k * seg_frames, (k + 1) * seg_frames) are a stream, in
    # order: the streams-mode tests and bench take S = n / seg_frames streams of T = seg_frames frames from this)
    seg = seg_frames or SEG_FRAMES
    complexities = complexities or ((3,) if kind in ("nsq", "chain_nsq") else (5, 7, 10) if kind in ("dd", "chain_dd") else (3, 5, 8, 10))
    # one directory per (kind, size, complexities, seed): ranks of a multi-GPU job ask for different seeds at the same time
    cache = os.path.join(cache or os.environ.get("CONCENTUS_SILK_CACHE", "/tmp/concentus_silk_corpus"),
                         "%s%s_%d_%s_%d%s%s" % (kind, "_v2" if kind.startswith("chain") else "", n, "-".join(map(str, complexities)), seed,
                                                "" if variant == "wb20" else "_" + variant, "" if seg == SEG_FRAMES else "_seg%d" % seg))
    done = os.path.join(cache, "done")
    if not os.path.exists(done):
        if under_profiler():
            # the capture runs worker interpreters (and, for a stale capture library, make / gcc): never from a process the
            # profiler has instrumented -- tools/round_profile*.sh build every corpus in a plain python3 step first
            raise RuntimeError("silk_corpus: %s is not built and this process runs under rocprofv3; build it first with "
                               "`python3 tests/silk_corpus.py %d %s`" % (cache, n, kind))
        # built in a directory of its own and renamed into place: two processes asking for the same corpus at the same
        # time each build a private copy, the first rename wins, the loser's copy is dropped
        final, cache = cache, "%s.tmp.%d" % (cache, os.getpid())
        shutil.rmtree(cache, ignore_errors=True)
        os.makedirs(cache)
        for f in _files(cache, kind, n, "w+").values():
            f.flush()
        jobs, row = [], 0
        while row < n:
            take = min(seg, n - row)
            k = len(jobs)
            jobs.append((cache, kind, seed + 7919 * k + {"nsq": 0, "dd": 104729, "lpc": 1299709, "pred": 15485863, "fpc": 32452843, "gains": 49979687, "shape": 67867967, "prefilter": 86028121, "pitch": 104395301, "bits": 160481183, "vad": 179424673, "chain_nsq": 122949823, "chain_dd": 141650939}[kind], complexities[k % len(complexities)],
                         row, take, n, variant))
            row += take
        workers = workers or max(1, min(len(jobs), len(os.sched_getaffinity(0)), 16))
        if workers == 1:
            for j in jobs:
                _capture_segment(j)
        else:
            import multiprocessing as mp
            # the workers are CPU-only children: no profiler / tool preloads in their environment
            saved = {k: os.environ.pop(k) for k in list(os.environ) if k == "LD_PRELOAD" or k.startswith(_TOOL_ENV)}
            try:
                pool = mp.get_context("spawn").Pool(workers)
                try:
                    for _ in pool.imap_unordered(_capture_segment, jobs):
                        pass
                finally:
                    pool.close()          # let the workers exit by themselves (terminate() = SIGTERM, which a
                    pool.join()           # profiler's signal handler reports as "Aborted")
            finally:
                os.environ.update(saved)
        open(os.path.join(cache, "done"), "w").write("ok\n")
        try:
            os.rename(cache, final)
        except OSError:
            shutil.rmtree(cache, ignore_errors=True)          # somebody else finished the same corpus first
        cache = final
    return _files(cache, kind, n, "r")


if __name__ == "__main__":
    import time
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    kind = sys.argv[2] if len(sys.argv) > 2 else "nsq"
    t0 = time.time()
    c = corpus(n, kind)
    print("%d %s records in %.1f s:" % (n, kind, time.time() - t0), {k: v.shape for k, v in c.items()})
    if kind == "nsq":
        st = np.asarray(c["nsq_in"][:, 24:28]).view(np.int32)[:, 0]
        print("signalType histogram (0 inactive, 1 unvoiced, 2 voiced):", np.bincount(st, minlength=3))

#!/usr/bin/env python3
"""Per-stage cycle shares of the frame kernel (diagnostic build with s_memtime stamps). GPU only."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import concentus_amd as ca

NAMES = ["load+dc_reject+rate", "silence", "preemph", "prefilter(pitch)", "transient", "mdct+energies+patch", "normalise",
         "tf_analysis", "coarse_energy", "tf_enc+spread+dynalloc+trim", "vbr", "allocation", "fine_energy", "PVQ",
         "finalise", "done+store",
         "pvq:theta", "pvq:exp_rotation", "pvq:presearch", "pvq:greedy", "pvq:encode_pulses", "pvq:band_setup", "pvq:other",
         "pitch:downsample | lane: band head", "pitch:search | lane: after pop / descend rest", "pitch:remove_doubling | lane: leaf bits2pulses", "pitch:decimate4 | lane: stage entry", "pitch:xcorr_coarse | lane: X -> column + recombine", "pitch:best_coarse | lane: time-divide haar", "pitch:xcorr_fine | lane: band tail + reconvergence", "front:mdct", "front:band_energies"]
NS = 32

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
kind = sys.argv[2] if len(sys.argv) > 2 else "noise"
rng = np.random.default_rng(3)
if kind == "noise":
    pcm = rng.integers(-8192, 8192, size=(n, 960, 2), dtype=np.int16)
else:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import encode_cases as ec
    pcm = ec.golden_module().synth_pcm("music", n, 4)
d = torch.from_numpy(pcm).cuda()
cfg = ca.default_config()
stride = ca.encoder.out_stride_for(cfg)
out = torch.zeros((n, stride), dtype=torch.uint8, device="cuda")
lens = torch.zeros(n, dtype=torch.int32, device="cuda")
rg = torch.zeros(n, dtype=torch.int32, device="cuda")
st = torch.zeros((4096, NS), dtype=torch.int64, device="cuda")
L = ca.lib.load()
D = ca.lib.load_diag()
ws = torch.empty((L.opusgpu_encode_workspace_bytes(n),), dtype=torch.uint8, device="cuda")
g = D.opusgpu_encode_batch_diag(C.byref(cfg), d.data_ptr(), out.data_ptr(), stride, lens.data_ptr(), rg.data_ptr(), n,
                                ws.data_ptr(), ws.numel(), st.data_ptr(), None)
torch.cuda.synchronize()
assert g == 0, g
tot = st.sum(0).cpu().numpy().astype(np.float64)
per_frame = tot / n
res = {NAMES[k]: round(per_frame[k]) for k in range(NS)}
res["_total_cycles_per_frame"] = round(per_frame.sum())
print(json.dumps(res, indent=1))
if len(sys.argv) > 3 and sys.argv[3] == "lane":
    # the back phase again, one lane per frame, on the same FrameMid records: cycles per wavefront (64 frames)
    st2 = torch.zeros((4096, NS), dtype=torch.int64, device="cuda")
    g = D.opusgpu_back_lane_diag(C.byref(cfg), ws.data_ptr(), out.data_ptr(), stride, lens.data_ptr(), rg.data_ptr(), n,
                                 st2.data_ptr(), None)
    torch.cuda.synchronize()
    assert g == 0, g
    waves = (n + 63) // 64
    per_wave = st2[:min(waves, 4096)].sum(0).cpu().numpy().astype(np.float64) / min(waves, 4096)
    print("lane-per-frame back phase, cycles per wavefront:")
    print(json.dumps({NAMES[k]: round(per_wave[k]) for k in range(NS) if per_wave[k] > 0}, indent=1))
    print("total", round(per_wave.sum()), "shares %:", {NAMES[k]: round(100 * per_wave[k] / per_wave.sum(), 1) for k in range(NS) if per_wave[k] > 0})
print("shares %:", {NAMES[k]: round(100 * per_frame[k] / per_frame.sum(), 1) for k in range(NS)})

#!/usr/bin/env python3
"""Per-stage cycles of the decoder's lane kernel (diagnostic build with s_memtime stamps), per wavefront. GPU only."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import concentus_amd as ca

NAMES = {0: "header", 1: "coarse_energy", 2: "tf+dynalloc+allocation+fine", 3: "pvq:ec_dec_uint", 4: "pvq:cwrsi", 5: "pvq:normalise",
         6: "pvq:rotation", 7: "band:setup(lowband haar/deinterleave)", 8: "partition walk + theta + fill", 9: "band:resynthesis",
         10: "quant_all_bands rest", 11: "finalise + anti-collapse + state", 12: "dual/stereo dispatch", 13: "stereo theta",
         14: "after band (masks, balance)", 15: "band head (tell, b, fold masks)", 16: "call -> quant_band entry", 17: "stereo: between the two quant_bands / before merge", 18: "stereo_merge"}
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
pcm = torch.from_numpy(np.random.default_rng(3).integers(-8192, 8192, size=(n, 960, 2), dtype=np.int16)).cuda()
pk, ln, _ = ca.encode_independent(pcm)
dec = ca.OpusDecoderBatch(n)
L = ca.lib.load()
D = ca.lib.load_diag()
st = torch.zeros((4096, 32), dtype=torch.int64, device="cuda")
ret = torch.zeros(n, dtype=torch.int32, device="cuda")
rng = torch.zeros(n, dtype=torch.int32, device="cuda")
rc = D.opusgpu_decode_lane_diag(dec._states.data_ptr(), pk.data_ptr(), pk.shape[1], ln.data_ptr(), ret.data_ptr(), rng.data_ptr(), n,
                                st.data_ptr(), None)
torch.cuda.synchronize()
assert rc == 0 and (ret.cpu().numpy() == 960).all()
w = min((n + 63) // 64, 4096)
per = st[:w].sum(0).cpu().numpy().astype(np.float64) / w
print(json.dumps({NAMES[k]: round(per[k]) for k in NAMES}, indent=1))
print("total", round(per.sum()), {NAMES[k]: round(100 * per[k] / per.sum(), 1) for k in NAMES})

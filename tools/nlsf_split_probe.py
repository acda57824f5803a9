#!/usr/bin/env python3
"""Where the time of silk_process_nlsfs_kernel goes: the same records with the survivor count forced to 1 / 4 / 32 (timing
only -- the outputs of the modified records are not the reference's). GPU only."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import concentus_amd as ca  # noqa: E402
import silk_corpus  # noqa: E402


def main():
    rec = silk_corpus.corpus(65536, "pred")
    base = np.array(rec["nlsf_in"])
    for surv in (0, 1, 4, 32):
        a = base.copy()
        if surv:
            a[:, 64 + 20:64 + 24].view(np.int32)[:, 0] = surv
        d = torch.from_numpy(a).cuda()
        o = torch.empty((a.shape[0], 120), dtype=torch.uint8, device="cuda")
        ca.silk_process_NLSFs(d, o)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            ca.silk_process_NLSFs(d, o)
        torch.cuda.synchronize()
        print("survivors", surv or "as captured", "%.3f ms" % ((time.perf_counter() - t0) / 3 * 1e3))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Experiment (GPU box): K = 6 / 7 scan kernels, old 16-bit form (FRISK_K8_BITS=16) against the narrow-counter form, on the
C2 shape (4.64 Mb x 8 copies for a longer launch) - needs a -DFRISK_TUNE library."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from frisk_amd import Engine, synth
for kmax, w, inc, lens in ((6, 5000, 500, synth.C2_LENS * 8), (7, 5000, 1000, [40_000_000]), (6, 2000, 500, [40_000_000])):
    with Engine(1, kmax) as e:
        e.synth(lens, seed=0xC2, island_frac=0.02, n_frac=0.0)
        e.profile_reset(); e.profile_add(); e.profile_finalize()
        ref = None
        for bits in ("16", "0"):
            os.environ["FRISK_K8_BITS"] = bits
            ts = []
            for _ in range(3):
                r = e.scan(w, inc, pinned=True); ts.append(e.kernel_ms(0))
            kld = r.kld[r.kept].copy()
            if ref is None: ref = kld
            print(json.dumps({"kmax": kmax, "w": w, "inc": inc, "form": "16-bit" if bits == "16" else "narrow", "candidates": int(r.n_candidates),
                              "scan_ms": min(ts), "Mwin_per_s": r.n_candidates / min(ts) / 1e3, "stat": e.scan_stat(),
                              "max_abs_diff": float(np.max(np.abs(kld - ref)))}), flush=True)

#!/usr/bin/env python3
"""Experiment (GPU box): time the K = 8 scan kernel forms on the bench shard and check that they agree.
Needs a -DFRISK_TUNE build of the library (FRISK_HIP_LIB=...), whose launcher reads FRISK_K8_BITS (8, 4, 0 = the r1
one-workgroup 16-bit form) from the environment at every scan."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from frisk_amd import Engine, synth  # noqa: E402
from frisk_amd import _ffi  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
variants = sys.argv[2].split(",") if len(sys.argv) > 2 else ["16", "8", "4", "0"]
CONFIGS = [
    ("C5/8 shard w=5000 i=1000", [max(1, int(x * scale)) for x in synth.c5_shard_lens(8, 0)], 5000, 1000, 0.07),
    ("C4 w=2000 i=500", [int(synth.C4_LENS[0] * scale)], 2000, 500, 0.07),
    ("C3 w=5000 i=1000", synth.C3_LENS, 5000, 1000, 0.001),
]
for name, lens, w, inc, nfrac in CONFIGS:
    with Engine(1, 8) as e:
        e.synth(lens, seed=0xC5, island_frac=0.02, n_frac=nfrac)
        e.profile_reset(); e.profile_add(); e.profile_finalize()
        ref = None
        for bits in variants:
            os.environ["FRISK_K8_BITS"] = bits
            ts = []
            for _ in range(3):
                r = e.scan(w, inc, pinned=True)
                ts.append(e.kernel_ms(0))
            k = r.kept
            kld = r.kld[k].copy()
            ovf = [int(_ffi.lib().frisk_last_scan_stat(e._ctx, i)) for i in range(3)]
            if ref is None:
                ref = kld
            print(json.dumps({"config": name, "bits": bits, "candidates": int(r.n_candidates), "rows": int(k.sum()),
                              "scan_ms": min(ts), "Mwin_per_s": r.n_candidates / min(ts) / 1e3,
                              "overflow": ovf, "kld_sum": float(kld.sum()),
                              "max_abs_diff_vs_first": float(np.max(np.abs(kld - ref))) if len(kld) == len(ref) else None}),
                  flush=True)

#!/usr/bin/env python3
"""Kernel-level timings of BASELINE.json's single-GPU-sized configurations on synthetic scaffolds of the named
shapes (one GPU; the multi-GPU configs are run on one GPU here just to time the kernels).  Prints one JSON line
per config."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frisk_amd import Engine, synth  # noqa: E402

CONFIGS = [
    ("C1 50 kb, k=1..4, w=5000 i=1000", [50000], 1, 4, 5000, 1000, 0.0),
    ("C2 E. coli shape 4.64 Mb, k=1..6, w=5000 i=500", synth.C2_LENS, 1, 6, 5000, 500, 0.0),
    ("C3 S. cerevisiae shape 12.07 Mb / 16 chr, k=1..8, w=5000 i=1000", synth.C3_LENS, 1, 8, 5000, 1000, 0.001),
    ("C4 human chr1 shape 249 Mb, k=1..8, w=2000 i=500", synth.C4_LENS, 1, 8, 2000, 500, 0.07),
    ("C5/8 shard 410 Mb, k=1..8, w=5000 i=1000", synth.c5_shard_lens(8, 0), 1, 8, 5000, 1000, 0.07),
    ("default geometry on the C5/8 shard, k=1..8, w=5000 i=2500", synth.c5_shard_lens(8, 0), 1, 8, 5000, 2500, 0.07),
]

for name, lens, kmin, kmax, w, inc, nfrac in CONFIGS:
    with Engine(kmin, kmax) as e:
        e.synth(lens, seed=17, island_frac=0.02, n_frac=nfrac)
        ts, tp = [], []
        for _ in range(3):
            e.profile_reset(); e.profile_add(); tp.append(e.kernel_ms(1)); e.profile_finalize()
            r = e.scan(w, inc, pinned=True)
            ts.append(e.kernel_ms(0))
        n = r.n_candidates
        print(json.dumps({"config": name, "bases": sum(lens), "candidates": n, "rows": int(r.kept.sum()),
                          "scan_ms": min(ts), "profile_ms": min(tp),
                          "scan_windows_per_s": n / (min(ts) * 1e-3) if min(ts) > 0 else None,
                          "profile_gbases_per_s": sum(lens) / (min(tp) * 1e-3) / 1e9 if min(tp) > 0 else None}), flush=True)

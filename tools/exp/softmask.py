#!/usr/bin/env python3
"""Experiment (GPU box): scan time of the bench shard with 0 % and 50 % soft-masked sequence (k=1..8 and k=2..8)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from frisk_amd import Engine, synth
lens = synth.c5_shard_lens(8, 0)
for kmin in (1, 2):
    for lf in (0.0, 0.5):
        with Engine(kmin, 8) as e:
            e.synth(lens, seed=0xC5, island_frac=0.02, n_frac=0.07, lower_frac=lf)
            e.profile_reset(); e.profile_add(); e.profile_finalize()
            ts = []
            for _ in range(4):
                r = e.scan(5000, 1000, pinned=True); ts.append(e.kernel_ms(0))
            print("kmin %d lower_frac %.1f: scan %.3f ms  kept %d  stat %s" % (kmin, lf, min(ts), int(r.kept.sum()), e.scan_stat()), flush=True)

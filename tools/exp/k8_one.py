#!/usr/bin/env python3
"""Experiment (GPU box): N scans of the bench shard (k=1..8 w=5000 i=1000) for profiling under rocprofv3.
usage: k8_one.py [scale] [n_scans]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from frisk_amd import Engine, synth  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
nscan = int(sys.argv[2]) if len(sys.argv) > 2 else 3
lens = [max(1, int(x * scale)) for x in synth.c5_shard_lens(8, 0)]
with Engine(1, 8) as e:
    e.synth(lens, seed=0xC5, island_frac=0.02, n_frac=0.07)
    e.profile_reset(); e.profile_add(); e.profile_finalize()
    for _ in range(nscan):
        r = e.scan(5000, 1000, pinned=True)
    print("scan_ms", e.kernel_ms(0), "candidates", r.n_candidates, "kld_sum", float(r.kld[r.kept].sum()))

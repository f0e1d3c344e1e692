#!/usr/bin/env python3
"""(GPU; run with FRISK_HIP_LIB = a -DFRISK_TUNE build and FRISK_K8_BITS=4 to take the sample out) first and second scan of a
freshly generated batch, scan kernels by HIP events: how much of a cold step's extra time is the sample, how much the fresh batch."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from frisk_amd.engine import Engine
from frisk_amd import synth
lens = synth.c5_shard_lens(8, 0)
kw = dict(seed=0xC5, island_frac=0.02, n_frac=0.07, lower_frac=0.0)
with Engine(1, 8) as e:
    for i in range(4):
        e.synth(lens, **kw)
        e.profile_reset(); e.profile_add(); e.profile_finalize()
        ts = []
        for _ in range(3):
            e.scan(5000, 1000, pinned=True); ts.append(round(e.kernel_ms(0), 3))
        print(json.dumps({"fresh_batch": i, "scan_ms_first_second_third": ts, "K8_BITS": os.environ.get("FRISK_K8_BITS"), "stat": e.scan_stat()}), flush=True)

#!/usr/bin/env python3
"""(GPU, under rocprofv3 --kernel-trace) three cold steps - batch regenerated, so the adaptive width samples - after two warm ones:
where the first scan of a batch spends its extra time.  Prints the HIP-event scan time of each."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from frisk_amd.engine import Engine
from frisk_amd import synth
lens = synth.c5_shard_lens(8, 0)
kw = dict(seed=0xC5, island_frac=0.02, n_frac=0.07, lower_frac=0.0)
with Engine(1, 8) as e:
    def step():
        e.profile_reset(); e.profile_add(); e.profile_finalize()
        t0 = time.perf_counter()
        r = e.scan(5000, 1000, pinned=True)
        return e.kernel_ms(0), (time.perf_counter() - t0) * 1e3
    e.synth(lens, **kw)
    print("first", step()); print("warm", step()); print("warm", step())
    for i in range(3):
        e.synth(lens, **kw)
        print("cold", step(), e.scan_stat(), flush=True)
    print("warm", step())

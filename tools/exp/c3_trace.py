#!/usr/bin/env python3
"""(GPU) the C3 shape (12 Mb, 16 scaffolds, 12 063 windows) scanned a few times: for `rocprofv3 --kernel-trace` - what a short scan's
launches look like on the timeline (tools/exp/c3_timeline.py prints the last scan's)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from frisk_amd import Engine, synth
with Engine(1, 8) as e:
    e.synth(synth.C3_LENS, seed=17, island_frac=0.02, n_frac=0.001)
    e.profile_reset(); e.profile_add(); e.profile_finalize()
    ts = []
    for _ in range(6):
        r = e.scan(5000, 1000, pinned=True); ts.append(e.kernel_ms(0))
    print(json.dumps({"scan_ms": [round(t, 4) for t in ts], "candidates": r.n_candidates, "stat": e.scan_stat()}))

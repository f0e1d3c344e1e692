#!/usr/bin/env python3
"""(GPU) C3 (12 063 windows) window by window (the default for so short a scan) against the schedule of a long scan (chunks of 8, tables sliding, ring)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from frisk_amd import Engine, synth
with Engine(1, 8) as e:
    e.synth(synth.C3_LENS, seed=17, island_frac=0.02, n_frac=0.001)
    e.profile_reset(); e.profile_add(); e.profile_finalize()
    for kw in (dict(), dict(chunks=True), dict(bits4=True), dict(chunks=True, bits4=True)):
        ts = []
        for _ in range(6):
            r = e.scan(5000, 1000, pinned=True, **kw); ts.append(e.kernel_ms(0))
        print(json.dumps({"kw": kw, "scan_ms_best": round(min(ts), 4), "M_windows_per_s": round(r.n_candidates / min(ts) / 1e3, 2), "stat": e.scan_stat(), "kld": float(r.kld[r.kept].sum())}))

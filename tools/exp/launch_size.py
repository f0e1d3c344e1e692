#!/usr/bin/env python3
"""(GPU) scan time against the number of windows of one launch (bench shard, K = 8, hinted 4-bit width): what a short launch - the
sample of the adaptive width, the tail segment - costs beyond its share of a long one."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from frisk_amd.engine import Engine
from frisk_amd import synth
lens = synth.c5_shard_lens(8, 0)
with Engine(1, 8) as e:
    e.synth(lens, seed=0xC5, island_frac=0.02, n_frac=0.07, lower_frac=0.0)
    e.profile_reset(); e.profile_add(); e.profile_finalize()
    r = e.scan(5000, 1000, pinned=True)
    for n in (768 * 16, 768 * 32, 768 * 48, 768 * 64, 768 * 128, 131000, 262144, 410610):
        ts = []
        for rep in range(5):
            e.scan(5000, 1000, pinned=True, c0=100000 if n < 300000 else 0, c1=(100000 if n < 300000 else 0) + n)
            ts.append(e.kernel_ms(0))
        print(json.dumps({"windows": n, "chunks_per_workgroup": n / 16 / 768, "scan_ms": round(min(ts), 3), "us_per_1000_windows": round(min(ts) / n * 1e6, 2), "stat": e.scan_stat()}), flush=True)

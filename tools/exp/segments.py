#!/usr/bin/env python3
"""(GPU, FRISK_TUNE build) the whole-C5 step (profile + scan + rows in page-locked host memory) with the scan in two row segments (the tail on a
second stream while the first rows travel) against one segment (FRISK_ONE_SEGMENT) and against a tail of any size (FRISK_TAIL_ANY)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, json, time
sys.path.insert(0, %r)
from frisk_amd import Engine, synth
lens = [n for r in range(8) for n in synth.c5_shard_lens(8, r)] if %r == "c5" else synth.c5_shard_lens(8, 0)
with Engine(1, 8) as e:
    e.synth(lens, seed=0xC5, island_frac=0.02, n_frac=0.07, lower_frac=0.0)
    def step():
        e.profile_reset(); e.profile_add(); e.profile_finalize()
        return e.scan(5000, 1000, pinned=True)
    for _ in range(3): step()
    ts, ks = [], []
    for _ in range(8):
        t0 = time.perf_counter(); step(); ts.append(time.perf_counter() - t0); ks.append(e.kernel_ms(0))
    print(json.dumps({"step_ms_best": round(min(ts) * 1e3, 3), "step_ms_median": round(sorted(ts)[4] * 1e3, 3), "scan_kernel_ms_best": round(min(ks), 3), "segments": e.scan_stat()[3]}))
'''
for shape in ("c5", "shard"):
    for rnd in range(2):
        for label, env in (("two segments", {}), ("one segment", {"FRISK_ONE_SEGMENT": "1"}), ("tail of any size", {"FRISK_TAIL_ANY": "1"})):
            o = subprocess.run([sys.executable, "-c", CHILD % (ROOT, shape)], env=dict(os.environ, FRISK_HIP_LIB=os.path.join(ROOT, "build/ab/tune.so"), **env), capture_output=True, text=True)
            print(shape, label, o.stdout.strip() or o.stderr[-300:], flush=True)

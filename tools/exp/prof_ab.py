#!/usr/bin/env python3
"""(GPU) profile kernel time (HIP events) of prebuilt libraries on the shard and the whole C5 shape: usage prof_ab.py lib1.so lib2.so ..."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, json
sys.path.insert(0, %r)
from frisk_amd import Engine, synth
out = {}
with Engine(1, 8) as e:
    for label, lens in (("shard", synth.c5_shard_lens(8, 0)), ("c5", [n for r in range(8) for n in synth.c5_shard_lens(8, r)])):
        e.synth(lens, seed=0xC5, island_frac=0.02, n_frac=0.07, lower_frac=0.0)
        ts = []
        for _ in range(6):
            e.profile_reset(); e.profile_add(); ts.append(e.kernel_ms(1)); e.profile_finalize()
        out[label] = {"profile_ms_best": round(min(ts), 4), "raw_sum": int(e.profile_raw().sum())}
print(json.dumps(out))
''' % ROOT
for rnd in range(2):
    for lib in sys.argv[1:]:
        o = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, FRISK_HIP_LIB=os.path.abspath(lib)), capture_output=True, text=True)
        print(os.path.basename(lib), o.stdout.strip() or o.stderr[-300:], flush=True)

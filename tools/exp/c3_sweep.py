#!/usr/bin/env python3
"""(GPU, FRISK_TUNE build) chunk length sweep on short scans: C3 (12 063 windows) and prefixes of it; FRISK_SCAN_CHUNK sets the chunk."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, json, os
sys.path.insert(0, %r)
from frisk_amd import Engine, synth
with Engine(1, 8) as e:
    e.synth(synth.C3_LENS, seed=17, island_frac=0.02, n_frac=0.001)
    e.profile_reset(); e.profile_add(); e.profile_finalize()
    out = {}
    for n in (12063, 6000, 3000, 1500):
        ts = []
        for _ in range(6):
            r = e.scan(5000, 1000, pinned=True, c0=0, c1=n, bits4=True); ts.append(e.kernel_ms(0))
        out[n] = round(min(ts) * 1e3, 1)
    print(json.dumps(out))
''' % ROOT
for chunk in ("1", "2", "4", "6", "8", "12", "16"):
    o = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, FRISK_HIP_LIB=os.path.join(ROOT, "build/ab/tune.so"), FRISK_SCAN_CHUNK=chunk), capture_output=True, text=True)
    print("chunk", chunk, "scan us by windows:", o.stdout.strip() or o.stderr[-300:], flush=True)

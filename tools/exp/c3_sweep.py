#!/usr/bin/env python3
"""(GPU, FRISK_TUNE build) chunk length sweep on short scans: the first n windows of the C5/8 shard; FRISK_SCAN_CHUNK sets the chunk, FRISK_NO_DEAL the static deal."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, json, os
sys.path.insert(0, %r)
from frisk_amd import Engine, synth
with Engine(1, 8) as e:
    e.synth(synth.c5_shard_lens(8, 0), seed=17, island_frac=0.02, n_frac=0.001)
    e.profile_reset(); e.profile_add(); e.profile_finalize()
    out = {}
    for n in (12063, 6000, 3000, 1500, 24000, 48000):
        ts = []
        for _ in range(6):
            r = e.scan(5000, 1000, pinned=True, c0=0, c1=n, bits4=True); ts.append(e.kernel_ms(0))
        out[n] = round(min(ts) * 1e3, 1)
    print(json.dumps(out))
''' % ROOT
for chunk in ("1", "2", "3", "4", "5", "6", "8", "12", "16"):
    for nodeal in ("", "1"):
        if nodeal and int(chunk) < 4:
            continue
        env = dict(os.environ, FRISK_HIP_LIB=os.path.join(ROOT, "build/ab/tune.so"), FRISK_SCAN_CHUNK=chunk)
        if nodeal:
            env["FRISK_NO_DEAL"] = "1"
        o = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        print("chunk", chunk, "static deal" if (nodeal or int(chunk) < 4) else "dealt by counters", "scan us by windows:", o.stdout.strip() or o.stderr[-300:], flush=True)

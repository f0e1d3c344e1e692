#!/usr/bin/env python3
"""(GPU) rows of prebuilt library variants against the first one: usage ab_kld.py base.so other.so ... - scans the bench shard and the
repeat-rich shard with each library and prints the largest absolute KLD difference to the first library's rows (and whether the other
columns are identical)."""
import os
import subprocess
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from frisk_amd import Engine, synth
e = Engine(1, 8)
out = {}
for label, kw in (("shard", dict(island_frac=0.02, n_frac=0.07, lower_frac=0.0)), ("mixed", synth.REPEATS_MIXED)):
    e.synth(synth.c5_shard_lens(8, 0), seed=0xC5, **kw)
    e.profile_reset(); e.profile_add(); e.profile_finalize()
    r = e.scan(5000, 1000)
    out[label + "_kld"] = r.kld.copy(); out[label + "_gc"] = r.gc.copy(); out[label + "_status"] = r.status.copy()
np.savez(%r, **out)
'''
libs = sys.argv[1:]
ref = None
for lib in libs:
    path = "/tmp/ab_kld_%s.npz" % os.path.basename(lib)
    run = subprocess.run([sys.executable, "-c", CHILD % (ROOT, path)], env=dict(os.environ, FRISK_HIP_LIB=os.path.abspath(lib)), capture_output=True, text=True)
    if run.returncode:
        print(lib, "FAILED", run.stderr[-400:]); continue
    d = np.load(path)
    if ref is None:
        ref = d; print(os.path.basename(lib), "reference"); continue
    for label in ("shard", "mixed"):
        k = (ref[label + "_status"] & 1) != 0
        a, b = ref[label + "_kld"][k], d[label + "_kld"][k]
        ok = np.isfinite(a) & np.isfinite(b)
        print(os.path.basename(lib), label, "rows", int(k.sum()), "max |dKLD| %.3e" % float(np.max(np.abs(a[ok] - b[ok]))), "mean dKLD %.3e" % float(np.mean(b[ok] - a[ok])),
              "status equal", bool(np.array_equal(ref[label + "_status"], d[label + "_status"])), "gc equal", bool(np.array_equal(ref[label + "_gc"], d[label + "_gc"], equal_nan=True)), flush=True)

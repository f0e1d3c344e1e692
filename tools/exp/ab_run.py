#!/usr/bin/env python3
"""(GPU) same-box A/B of prebuilt libraries: usage ab_run.py [rounds] lib1.so lib2.so ... - for each library in turn, `rounds` times over:
HIP-event time of the scan kernels on the bench shard (410 Mb) and on the whole C5 shape (3.29 Gb), best of 6, plus a checksum of the rows."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, json
sys.path.insert(0, %r)
from frisk_amd import Engine, synth
out = {}
e = Engine(1, 8)
for label, lens, kw in (("shard", synth.c5_shard_lens(8, 0), dict(island_frac=0.02, n_frac=0.07, lower_frac=0.0)),
                        ("mixed", synth.c5_shard_lens(8, 0), synth.REPEATS_MIXED),
                        ("c5", [n for r in range(8) for n in synth.c5_shard_lens(8, r)], dict(island_frac=0.02, n_frac=0.07, lower_frac=0.0))):
    if label in %r: continue
    e.synth(lens, seed=0xC5, **kw)
    e.profile_reset(); e.profile_add(); e.profile_finalize()
    ts = []
    for _ in range(6):
        r = e.scan(5000, 1000, pinned=True); ts.append(e.kernel_ms(0))
    out[label] = {"best": round(min(ts), 4), "first": round(ts[0], 3), "last": round(ts[-1], 3), "kld_sum": float(r.kld[r.kept].sum()), "stat": e.scan_stat()[:3]}
print(json.dumps(out))
'''
args = sys.argv[1:]
rounds = int(args.pop(0)) if args and args[0].isdigit() else 2
skip = os.environ.get("AB_SKIP", "")
for rnd in range(rounds):
    for lib in args:
        out = subprocess.run([sys.executable, "-c", CHILD % (ROOT, skip)], env=dict(os.environ, FRISK_HIP_LIB=os.path.abspath(lib)), capture_output=True, text=True)
        print(os.path.basename(lib), out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-400:], flush=True)

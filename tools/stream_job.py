#!/usr/bin/env python3
"""One whole C5-shaped job from page-locked host memory in the 0.25 B/base form, a few times over - the thing to put under
`rocprofv3 --kernel-trace --memory-copy-trace` to SEE the upload pieces and phase A's kernels side by side
(tools/profile_round.sh; summary in profiles/r4_*_stream_overlap.txt).  Prints one JSON line with the wall time per job."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from frisk_amd import Engine, synth  # noqa: E402

W, INC = 5000, 1000
jobs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
lens = [n for r in range(8) for n in synth.c5_shard_lens(8, r)]
with Engine(1, 8) as e:
    e.synth(lens, seed=0xC5, island_frac=0.02, n_frac=0.07, lower_frac=0.0)
    codes, inv_runs, low_runs = e.export_2bit(pinned=True)

    def job():
        e.stage_2bit(codes, inv_runs, low_runs, lens)
        e.commit()
        e.profile_reset()
        e.profile_add()
        e.profile_finalize()
        return e.scan(W, INC, pinned=True)
    job()
    t0 = time.perf_counter()
    for _ in range(jobs):
        res = job()
    dt = (time.perf_counter() - t0) / jobs
    rows = int(res.kept.sum())
    print(json.dumps({"jobs": jobs, "ms_per_job": dt * 1e3, "rows": rows, "windows_per_s": rows / dt,
                      "pcie_bytes_per_job": int(codes.nbytes + inv_runs.nbytes + low_runs.nbytes)}))

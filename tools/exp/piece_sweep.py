#!/usr/bin/env python3
"""(GPU) the streamed whole-C5 job (tools/stream_job.py) against the size of the upload pieces (frisk_seq_stage_2bit's piece_bases)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from frisk_amd import Engine, synth
lens = [n for r in range(8) for n in synth.c5_shard_lens(8, r)]
with Engine(1, 8) as e:
    e.synth(lens, seed=0xC5, island_frac=0.02, n_frac=0.07, lower_frac=0.0)
    codes, inv_runs, low_runs = e.export_2bit(pinned=True)
    for rnd in range(2):
        for mb in (8, 16, 32, 64, 128):
            def job():
                e.stage_2bit(codes, inv_runs, low_runs, lens, piece_bases=mb * 4 * (1 << 20))
                e.commit(); e.profile_reset(); e.profile_add(); e.profile_finalize()
                return e.scan(5000, 1000, pinned=True)
            job()
            t0 = time.perf_counter()
            for _ in range(4):
                res = job()
            dt = (time.perf_counter() - t0) / 4
            print(json.dumps({"piece_MB": mb, "ms_per_job": round(dt * 1e3, 2), "M_windows_per_s": round(int(res.kept.sum()) / dt / 1e6, 2)}), flush=True)

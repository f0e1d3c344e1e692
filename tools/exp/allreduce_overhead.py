#!/usr/bin/env python3
"""(GPU) what the in-stream RCCL all-reduce adds to a step: a one-rank nccl group (the collective still runs: force=True), bench shard,
steps with and without it, 30 each."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch, torch.distributed as dist
from frisk_amd.engine import Engine
from frisk_amd import synth
dist.init_process_group("nccl", rank=0, world_size=1)
lens = synth.c5_shard_lens(8, 0)
with Engine(1, 8) as e:
    e.synth(lens, seed=0xC5, island_frac=0.02, n_frac=0.07, lower_frac=0.0)
    def step(ar):
        e.profile_reset(); e.profile_add()
        if ar: e.profile_allreduce(force=True)
        e.profile_finalize()
        return e.scan(5000, 1000, pinned=True)
    for ar in (False, True, False, True):
        for _ in range(5): step(ar)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): step(ar)
        torch.cuda.synchronize()
        print(json.dumps({"allreduce": ar, "ms_per_step": round((time.perf_counter() - t0) / 30 * 1e3, 4)}), flush=True)
dist.destroy_process_group()

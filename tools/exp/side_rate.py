#!/usr/bin/env python3
"""(GPU) the bench shard in three shapes x the forms of the K = 8 bulk launch (default = what the sample picks, plain 4-bit,
4-bit + side table, 8-bit via FRISK_K8_BITS in -DFRISK_TUNE builds): scan kernel ms (HIP events), best of 5."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from frisk_amd.engine import Engine
from frisk_amd import synth
lens = synth.c5_shard_lens(8, 0)
clean = dict(seed=0xC5, island_frac=0.02, n_frac=0.07, lower_frac=0.0)
shapes = (("clean", clean), ("repeats unmasked", dict(synth.REPEATS_UNMASKED, seed=0xC5)), ("repeats soft-masked", dict(synth.REPEATS_SOFT, seed=0xC5)))
with Engine(1, 8) as e:
    for name, kw in shapes:
        e.synth(lens, **kw)
        e.profile_reset(); e.profile_add(); e.profile_finalize()
        ref = None
        for form, fk in (("default", {}), ("plain4", dict(bits4=True)), ("side4", dict(side4=True))):
            ts = []
            for _ in range(5):
                r = e.scan(5000, 1000, pinned=True, **fk)
                ts.append(e.kernel_ms(0))
            if ref is None: ref = r.kld.copy()
            same = bool((ref.view('u8') == r.kld.view('u8')).all())
            print(json.dumps({"shape": name, "form": form, "scan_ms": round(min(ts), 3), "stat": e.scan_stat(), "side": e.scan_side(), "same_bits": same}), flush=True)

import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from frisk_amd.engine import Engine
import test_gpu_slide as T
rng = np.random.default_rng(7300)
for case_no in range(8):
    c = T._case(rng)
    if case_no < 7:
        # consume what the test consumes from the stream
        with Engine(c["kmin"], c["kmax"]) as e:
            e.load(c["seqs"]); e.profile_reset(); e.profile_add(); e.profile_finalize()
            n = len(e.scan(c["w"], c["inc"], rip=c["rip"], scaffolds_all=c["scaffolds_all"]))
        if n > 12: rng.integers(1, 8)
        continue
    print(c["kmin"], c["kmax"], c["w"], c["inc"], c["scaffolds_all"], [len(s) for s in c["seqs"]])
    with Engine(c["kmin"], c["kmax"]) as e:
        e.load(c["seqs"]); e.profile_reset(); e.profile_add(); e.profile_finalize()
        for kw in (dict(), dict(chunks=True), dict(chunks=True, bits4=True), dict(chunks=True, side4=True)):
            r = e.scan(c["w"], c["inc"], rip=c["rip"], scaffolds_all=c["scaffolds_all"], **kw)
            print(kw, len(r), [int(e._lib.frisk_last_scan_stat(e._ctx, i)) for i in range(5)], e.scan_plan(c["w"], c["inc"], c["scaffolds_all"]))

import sys, numpy as np
sys.path.insert(0, '.')
from frisk_amd.engine import Engine
rng = np.random.default_rng(1)
s = rng.choice(np.frombuffer(b"ATGC", dtype=np.uint8), size=120000)
s[3000:3040] = ord('A'); s[9000:9060] = np.resize(np.frombuffer(b"CA", dtype=np.uint8), 60)
with Engine(2, 8) as e:
    e.load([s.tobytes()])
    e.profile_reset(); e.profile_add(); e.profile_finalize()
    a = e.scan(5000, 1000, chunks=True)
    print('default', e.scan_stat(), e.scan_side())
    b = e.scan(5000, 1000, chunks=True, bits4=True)
    print('bits4', e.scan_stat(), e.scan_side())
    c = e.scan(5000, 1000, chunks=True, side4=True)
    print('side4', e.scan_stat(), e.scan_side(), [int(e._lib.frisk_last_scan_stat(e._ctx, i)) for i in range(6)])
    print(np.array_equal(a.kld, b.kld, equal_nan=True), np.array_equal(a.kld, c.kld, equal_nan=True), np.nanmax(np.abs(a.kld - c.kld)))

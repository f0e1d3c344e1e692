#!/usr/bin/env python3
"""Extended differential fuzz of the scan kernels against the compiled C oracle (GPU box): like tests/test_gpu_fuzz.py but
biased towards what round 2 added - K = 6..8 with kmin <= K-3, windows <= 5120, low-complexity runs that wrap 4- and 8-bit
counters, many short invalid runs (orphan-list overflow), rescued small scaffolds longer than the kernel's reach, tiles.
FUZZ_R4=1 (round 4): the batch goes up in the 2-bit + run-list form in small pieces (stage_2bit, the profile following the
pieces), the profile takes the one-pass 16-bit kernel on every third seed, and every second seed scans with the job's own
schedule (no forced chunks: the device-side width verdict, the short-scan rounds and the packed row block decide).
usage: fuzz_long.py <first seed> <n seeds> [minutes]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from frisk_amd import _ffi
from frisk_amd.engine import Engine
from oracle import frisk_oracle_c as OC

seed0, nseeds = int(sys.argv[1]), int(sys.argv[2])
budget = float(sys.argv[3]) * 60 if len(sys.argv) > 3 else 1e9
t0 = time.time()
units = [b"A", b"C", b"AT", b"CAG", b"GATA", b"AAAT", b"TTAGGG", b"ACGTACGA"]
bad = 0
sided = handed_side = 0          # scans whose bulk launch ran the side-table form; windows they still handed on
for seed in range(seed0, seed0 + nseeds):
    if time.time() - t0 > budget:
        break
    rng = np.random.default_rng(seed)
    kmax = int(rng.choice([6, 7, 8, 8, 8]))
    kmin = int(rng.integers(1, kmax - 2))
    w = int(rng.choice([600, 1000, 2000, 2048, 2049, 3000, 5000, 5120]))
    inc = max(1, int(w * rng.choice([0.1, 0.2, 0.5, 1.0, 1.3])))
    big = os.environ.get("FUZZ_BIG") == "1"         # jobs of > 1024 windows: the adaptive sample and the 4-bit tier run
    if big:
        inc = max(1, w // 10)
    seqs = []
    for q in range(int(rng.integers(1, 5))):
        n = int(rng.choice([w // 2, w + 1, int(1.6 * w), 3 * w + 7, 12 * w + int(rng.integers(0, w)), 40 * w]))
        if big and q == 0:
            n = int(rng.integers(150, 400)) * w
        s = rng.choice(np.frombuffer(b"ATGC", dtype=np.uint8), size=n, p=rng.dirichlet([3, 3, 3, 3]))
        for _ in range(int(rng.integers(0, 12)) * (n // (40 * w) + 1)):
            a = int(rng.integers(0, max(1, n - 10)))
            kind = int(rng.integers(0, 5))
            if kind == 0:                                   # a tandem repeat: wraps 4-bit (>= 16+K-1 bases) or 8-bit counters
                u = units[int(rng.integers(0, len(units)))]
                ln = int(rng.choice([20, 40, 100, 300, 700]))
                rep = (u * (ln // len(u) + 1))[:ln]
                s[a:a + ln] = np.frombuffer(rep, dtype=np.uint8)[:len(s[a:a + ln])]
            elif kind == 1:                                 # sprinkled invalid bases: many orphans
                for p in range(a, min(n, a + int(rng.choice([200, 1500]))), int(rng.choice([9, 13, 40]))):
                    s[p] = ord("N")
            elif kind == 2:
                s[a:a + int(rng.choice([1, 8, 50, w // 4]))] = ord("N")
            elif kind == 3:
                s[a:a + int(rng.choice([5, 64, w // 3]))] |= 0x20
            else:
                s[a:a + 3] = np.frombuffer(b"RYK", dtype=np.uint8)[:len(s[a:a + 3])]
        seqs.append(s.tobytes())
    all_ = bool(rng.integers(0, 2))
    rip = bool(rng.integers(0, 2)) and kmin <= 2
    tag = "seed %d k=%d..%d w=%d i=%d all=%s lens=%s" % (seed, kmin, kmax, w, inc, all_, [len(x) for x in seqs])
    r4 = os.environ.get("FUZZ_R4") == "1"
    with Engine(kmin, kmax) as e:
        if r4:
            codes, ir, lr, lens = e.pack_2bit(seqs, pinned=bool(seed & 4))
            e.stage_2bit(codes, ir, lr, lens, piece_bases=int(rng.choice([0, 1024, 4096, 1 << 16])))
            e.commit()
            e.profile_reset(); e.profile_add(one_pass=seed % 3 == 0); e.profile_finalize()
        else:
            e.load(seqs)
            e.profile_reset(); e.profile_add(); e.profile_finalize()
        sym, tl, ex, nn = e.profile_get()
        # (round 3: the schedule of a long scan - chunks, sliding tables, the ring - on every case; 4-bit counters first on odd seeds)
        # (4-bit counters with the side table for period-4 max-mers on seeds = 3 mod 4; the rows must then equal the default scan's bit for bit)
        side = (seed & 3) == 3 and kmax == 8
        res = e.scan(w, inc, rip=rip, scaffolds_all=all_, chunks=not (r4 and seed & 2), bits4=bool(seed & 1) and kmax == 8 and not side, side4=side)
        stat = e.scan_stat() + (e.scan_side(),)
        sided += e.scan_side(); handed_side += stat[1] if e.scan_side() else 0
        if side:
            ref = e.scan(w, inc, rip=rip, scaffolds_all=all_)
            same = all(np.array_equal(getattr(res, c).view(np.uint64), getattr(ref, c).view(np.uint64)) for c in ("kld", "gc")) and \
                np.array_equal(res.status, ref.status)
            if not same:
                bad += 1
                print("BITS DIFFER (side table form against the default)", seed, flush=True)
    osym, ometa = OC.genome_profile(seqs, kmin, kmax)
    ok = np.array_equal(sym, osym) and (tl, ex, nn) == tuple(ometa)
    exp = OC.scan(seqs, OC.genome_ivom(osym, ometa, kmin, kmax), kmin, kmax, w, inc, scaffolds_all=all_, rip=rip)
    k = np.nonzero(res.kept)[0]
    ok = ok and len(k) == len(exp["kld"])
    if ok and len(k):
        zero = (exp["status"] & OC.ROW_ZERO_DIV) != 0
        ok = (np.array_equal(res.start[k], exp["start"]) and np.array_equal(res.gc[k], exp["gc"], equal_nan=True)
              and np.array_equal((res.status[k] & _ffi.ROW_ZERO_WEIGHT) != 0, zero)
              and np.array_equal((res.status[k] & _ffi.ROW_NO_MAXMER) != 0, (exp["status"] & OC.ROW_NO_MAXMER) != 0))
        if ok and rip:
            ok = all(np.array_equal(getattr(res, c)[k], exp[c], equal_nan=True) for c in ("pi", "si", "cri"))
        if ok and (~zero).any():
            ok = float(np.max(np.abs(res.kld[k][~zero] - exp["kld"][~zero]))) <= 1e-11
    if not ok:
        bad += 1
        print("MISMATCH", tag, stat, flush=True)
    elif seed % 20 == 0:
        print("ok", tag, "rows", len(k), "stat", stat, flush=True)
print("done: %d seeds, %d mismatches, %.0f s; %d scans in the side-table form (%d windows handed on by them)" % (
    seed - seed0 + 1, bad, time.time() - t0, sided, handed_side), flush=True)

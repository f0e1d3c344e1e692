"""GPU parity beyond the golden fixtures: the HIP path (through the C ABI) against the numpy oracle on
seeded synthetic scaffolds at sizes the oracle finishes in seconds, plus size-independent properties at
BASELINE.json's full single-GPU sizes.  Integer work bit-exact; KLD within KLD_TOL (north star: 1e-6)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KLD_TOL = 1e-10


def make_engine(kmin, kmax):
    from frisk_amd import Engine
    return Engine(kmin, kmax)


def oracle_rows(seqs, kmin, kmax, w, inc, mask_host=False, scaffolds_all=False, rip=False):
    from oracle import frisk_oracle_np as N
    prof = N.genome_profile(seqs, kmin, kmax, mask_host=mask_host)
    rows = N.scan([(str(i), s) for i, s in enumerate(seqs)], prof, kmin, kmax, w, inc, scaffolds_all, rip)
    return prof, rows


def synth_seqs(lens, seed, **kw):
    from frisk_amd import synth
    return [synth.scaffold(n, seed, i, **kw) for i, n in enumerate(lens)]


@pytest.mark.parametrize("kw", [dict(island_frac=0.4, n_frac=0.25, lower_frac=0.3),
                                dict(island_frac=0.1, n_frac=0.05, lower_frac=0.3, repeats_per_kb=1.2),      # soft-masked, with repeats
                                dict(island_frac=0.1, n_frac=0.05, lower_frac=0.0, repeats_per_kb=1.9),      # unmasked, with repeats
                                dict(island_frac=0.1, n_frac=0.05, lower_frac=0.0, repeats_per_kb=1.9, period_mix=0.5, sat_frac=0.3),
                                dict(island_frac=0.1, n_frac=0.02, lower_frac=0.2, repeats_per_kb=1.0, period_mix=1.0, sat_frac=0.2)])
def test_device_generator_equals_host_generator(kw):
    lens = [10000, 4096, 4097, 1, 33000, 70001] + ([1_400_000] if kw.get("sat_frac") else [])     # (satellite arrays: units of 131 072 bases)
    with make_engine(1, 4) as e:
        e.synth(lens, seed=99, **kw)
        host = synth_seqs(lens, 99, **kw)
        for i in range(len(lens)):
            assert e.read_seq(i) == host[i], "scaffold %d differs" % i
        assert e.read_seq(4, 100, 50) == host[4][100:150]


def test_pack_roundtrip_all_bytes():
    seq = bytes(range(1, 256)) * 3 + b"ACGTacgtNnRYKM-*"
    with make_engine(1, 3) as e:
        e.load([seq, b"", b"A"])
        back = e.read_seq(0)
        expect = bytes(c if c in b"ACGTacgt" else ord("N") for c in seq)
        assert back == expect
        assert e.read_seq(1) == b"" and e.read_seq(2) == b"A"


@pytest.mark.parametrize("kmin,kmax,w,inc,mask_host,rip", [
    (1, 8, 5000, 1000, False, True),      # the metric's geometry
    (1, 6, 5000, 500, False, True),       # C2
    (1, 8, 2000, 500, False, False),      # C4
    (2, 7, 3000, 700, True, True),
    (3, 5, 777, 100, False, False),
    (8, 8, 5000, 2500, False, False),
    (1, 1, 600, 600, False, False),
    (7, 8, 3000, 1500, False, False),     # no small tables at all (orders 7, 8 only)
    (6, 8, 3000, 1500, False, False),     # small table = order 6 only, no shared prefix
    (5, 7, 2000, 1000, False, False),     # shared prefix = the lowest order
    (6, 6, 1500, 500, False, False),
    (2, 2, 800, 400, False, True),
    (1, 8, 12000, 4000, False, True),     # longer than the unrolled fast paths: generic kernel
    (1, 6, 30000, 10000, False, False),
])
def test_scan_matches_numpy_oracle(kmin, kmax, w, inc, mask_host, rip):
    from frisk_amd import _ffi
    lens = [150000, 61234, 5000, 8751, 300]
    seqs = synth_seqs(lens, 7, island_frac=0.15, n_frac=0.12, lower_frac=0.08)
    (sym, meta), rows = oracle_rows(seqs, kmin, kmax, w, inc, mask_host=mask_host, rip=rip and kmin <= 2 <= kmax)
    with make_engine(kmin, kmax) as e:
        e.load(seqs)
        e.profile_reset()
        e.profile_add(mask_host=mask_host)
        e.profile_finalize()
        gsym, tl, ex, nn = e.profile_get()
        assert np.array_equal(gsym, sym) and (tl, ex, nn) == tuple(meta)
        res = e.scan(w, inc, rip=rip and kmin <= 2 <= kmax)
        kept = np.nonzero(res.kept)[0]
        assert len(kept) == len(rows) and len(rows) > 10
        worst = 0.0
        for r, exp in zip(kept.tolist(), rows):
            assert (str(res.seq_index[r]), int(res.start[r]), int(res.stop[r])) == (exp["name"], exp["start"], exp["stop"])
            if "error" in exp:
                assert res.status[r] & _ffi.ROW_ZERO_WEIGHT
                continue
            assert not (res.status[r] & _ffi.ROW_ZERO_WEIGHT)
            worst = max(worst, abs(float(res.kld[r]) - exp["KLD"]))
            assert float(res.gc[r]) == exp["GC"]
            if rip and kmin <= 2 <= kmax:
                got = (float(res.pi[r]), float(res.si[r]), float(res.cri[r]))
                assert all((a != a and b != b) or a == b for a, b in zip(got, exp["RIP"]))
        assert worst <= KLD_TOL, worst
        # bit-exact per-window counts on a sample of candidates (the full dump is large at K=8)
        c0 = int(kept[len(kept) // 3])
        dbg = e.scan(w, inc, c0=c0, c1=min(res.n_candidates, c0 + 12), debug=True)
        cand_to_row = {int(c): i for i, c in enumerate(kept.tolist())}
        checked = 0
        for t in range(len(dbg)):
            if dbg.kept[t]:
                exp = rows[cand_to_row[c0 + t]]
                assert np.array_equal(dbg.counts[t].astype(np.int64), exp["counts"])
                assert dbg.meta[t].tolist() == exp["meta"]
                assert dbg.kld[t] == res.kld[c0 + t]            # debug build of the kernel: same bits
                checked += 1
        assert checked > 0


def test_many_orphans_and_big_counts():
    """The scan kernel's slow paths: > 4 orphan 7-mers per window (many invalid runs) and counts outside the
    p table (low-complexity sequence), against the numpy oracle."""
    from frisk_amd import synth
    base = bytearray(synth.scaffold(30000, 21, 0, island_frac=0.3))
    for pos in range(40, 30000, 53):            # an N every 53 bases: ~95 orphans per 5 kb window, < 30 % N
        base[pos] = ord("N")
    low = bytearray(synth.scaffold(24000, 22, 0))
    low[3000:9000] = b"A" * 6000                # poly-A and a dinucleotide repeat: counts of thousands
    low[12000:16000] = b"AC" * 2000
    low[7000] = ord("n")
    seqs = [bytes(base), bytes(low)]
    for kmin, kmax, w, inc in ((1, 8, 5000, 1000), (1, 8, 2000, 500), (1, 7, 3000, 1000)):
        (sym, meta), rows = oracle_rows(seqs, kmin, kmax, w, inc)
        with make_engine(kmin, kmax) as e:
            e.load(seqs)
            e.profile_reset(); e.profile_add(); e.profile_finalize()
            res = e.scan(w, inc, debug=True)
            kept = np.nonzero(res.kept)[0]
            assert len(kept) == len(rows) > 10
            for r, exp in zip(kept.tolist(), rows):
                assert np.array_equal(res.counts[r].astype(np.int64), exp["counts"])
                assert abs(float(res.kld[r]) - exp["KLD"]) <= KLD_TOL
            fast = e.scan(w, inc)
            assert np.array_equal(fast.kld, res.kld, equal_nan=True)        # non-debug build: same bits


def test_profile_is_linear_over_ranges_and_batches():
    lens = [50000, 20011, 999]
    seqs = synth_seqs(lens, 11, island_frac=0.1, n_frac=0.1, lower_frac=0.1)
    for kmin, kmax in ((1, 8), (2, 5)):
        with make_engine(kmin, kmax) as e:
            e.load(seqs)
            e.profile_reset()
            e.profile_add()
            whole = e.profile_raw()
            e.profile_reset()
            P = e.padded_len
            cuts = [0, 1, 4097, P // 3, P // 2 + 5, P]
            for a, b in zip(cuts[:-1], cuts[1:]):
                e.profile_add(pos_begin=a, pos_end=b)
            assert np.array_equal(e.profile_raw(), whole)
            # two batches accumulate: load scaffolds separately
            e.profile_reset()
            for s in seqs:
                e.load([s])
                e.profile_add()
            assert np.array_equal(e.profile_raw(), whole)
            # finalised profile: strand-symmetric, and marginals consistent with the oracle
            e.profile_finalize()
            sym, tl, ex, nn = e.profile_get()
            from oracle import frisk_oracle_np as N
            osym, ometa = N.genome_profile(seqs, kmin, kmax)
            assert np.array_equal(sym, osym) and (tl, ex, nn) == tuple(ometa)
            for x in range(kmin, kmax + 1):
                o = N.table_offset(kmin, x)
                t = sym[o:o + 4 ** x]
                assert np.array_equal(t, t[N.revcomp_index(x)])


def test_scan_ranges_concatenate_and_runs_are_deterministic():
    lens = [120000, 33333]
    seqs = synth_seqs(lens, 5, island_frac=0.1, n_frac=0.05)
    with make_engine(1, 8) as e:
        e.load(seqs)
        e.profile_reset(); e.profile_add(); e.profile_finalize()
        full = e.scan(5000, 1000, rip=True)
        again = e.scan(5000, 1000, rip=True)
        for f in ("start", "stop", "status", "kld", "gc", "pi", "si", "cri"):
            assert np.array_equal(getattr(full, f), getattr(again, f), equal_nan=True), f
        n = full.n_candidates
        cuts = [0, 1, 7, n // 2, n - 3, n]
        parts = [e.scan(5000, 1000, rip=True, c0=a, c1=b) for a, b in zip(cuts[:-1], cuts[1:])]
        for f in ("seq_index", "start", "stop", "status", "kld", "gc", "cri"):
            cat = np.concatenate([getattr(p, f) for p in parts])
            assert np.array_equal(cat, getattr(full, f), equal_nan=True), f
        # a profile installed from a cache gives the same scores as the computed one
        sym, tl, ex, nn = e.profile_get()
        e.profile_reset()
        e.profile_set(sym, tl, ex, nn)
        cached = e.scan(5000, 1000, rip=True)
        assert np.array_equal(cached.kld, full.kld, equal_nan=True)


def test_error_paths_and_edge_batches():
    from frisk_amd import _ffi
    with make_engine(3, 5) as e:
        with pytest.raises(_ffi.FriskHipError) as ei:
            e.scan_plan(100, 10)
        assert ei.value.code == _ffi.E_STATE
        e.load([b"ACGT" * 100])
        with pytest.raises(_ffi.FriskHipError) as ei:
            e.scan(100, 10)
        assert ei.value.code == _ffi.E_STATE               # profile not finalised
        e.profile_reset(); e.profile_add(); e.profile_finalize()
        with pytest.raises(_ffi.FriskHipError) as ei:
            e.scan(100, 10, rip=True)                        # kmin > 2: no dinucleotide table (reference L478)
        assert ei.value.code == _ffi.E_ARG
        assert len(e.scan(70000, 10)) == 0                   # long windows are supported (global-memory path): no candidates here
        with pytest.raises(_ffi.FriskHipError) as ei:
            e.scan(0, 10)
        assert ei.value.code == _ffi.E_ARG
        with pytest.raises(_ffi.FriskHipError):
            e.profile_add(pos_begin=5, pos_end=10 ** 9)
    with pytest.raises(_ffi.FriskHipError):
        make_engine(1, 13)
    with pytest.raises(_ffi.FriskHipError):
        make_engine(0, 3)
    with make_engine(1, 8) as e:
        e.load([])                                            # empty batch
        e.profile_reset(); e.profile_add(); e.profile_finalize()
        sym, tl, ex, nn = e.profile_get()
        assert sym.sum() == 0 and (tl, ex, nn) == (0, 0, 0)
        assert len(e.scan(5000, 1000)) == 0
        e.load([b"", b"N" * 9000, b"acgt" * 3000, b"ACGTTGCA" * 2000])
        e.profile_reset(); e.profile_add(); e.profile_finalize()
        res = e.scan(5000, 1000)
        # all-N and all-lowercase scaffolds: every window is dropped (lowercase counts as N for the filter)
        assert res.kept.sum() == 16 and set(res.seq_index[res.kept].tolist()) == {3}
        assert np.all(res.kld[res.kept] > -1e-12)


@pytest.mark.parametrize("shape", ["C1", "C2", "C3", "C4", "C5"])
def test_full_size_properties(shape):
    """BASELINE configs at full size: properties that need no oracle run."""
    from frisk_amd import synth
    if shape == "C1":                   # BASELINE configs[0]: single 50 kb scaffold, k=1..4, w=5000 i=1000
        lens, kmin, kmax, w, inc, nfrac = [50000], 1, 4, 5000, 1000, 0.0
    elif shape == "C2":
        lens, kmin, kmax, w, inc, nfrac = synth.C2_LENS, 1, 6, 5000, 500, 0.0
    elif shape == "C3":
        lens, kmin, kmax, w, inc, nfrac = synth.C3_LENS, 1, 8, 5000, 1000, 0.001
    elif shape == "C4":                 # one 249 Mb scaffold
        lens, kmin, kmax, w, inc, nfrac = synth.C4_LENS, 1, 8, 2000, 500, 0.07
    else:                               # the whole 3.3 Gb GRCh38-shaped assembly on one GPU: 3.28 M candidate windows
        lens = [n for r in range(8) for n in synth.c5_shard_lens(8, r)]
        kmin, kmax, w, inc, nfrac = 1, 8, 5000, 1000, 0.07
    with make_engine(kmin, kmax) as e:
        e.synth(lens, seed={"C1": 1, "C2": 2, "C3": 3, "C4": 4, "C5": 5}[shape], island_frac=0.02, n_frac=nfrac)
        e.profile_reset(); e.profile_add(); e.profile_finalize()
        sym, tl, ex, nn = e.profile_get()
        assert tl == sum(lens)
        top = sym[-(4 ** kmax):]
        assert top.sum() == 2 * (sum(max(0, n - kmax + 1) for n in lens) - ex)      # every counted word twice (L350-351)
        res = e.scan(w, inc)
        assert res.n_candidates == sum(n // inc for n in lens)                      # floor(size/i) per scaffold (L228)
        k = res.kept
        assert np.all(np.isfinite(res.kld[k])) and np.all(res.kld[k] > -1e-12)      # a KL divergence is >= 0
        assert np.all((res.gc[k] >= 0) & (res.gc[k] <= 1))
        assert not np.any(res.zero_weight)
        # rows come out in scaffold order, starts ascending; jumpback rows of a scaffold are identical duplicates
        assert np.all(np.diff(res.seq_index) >= 0)
        from frisk_amd import _ffi
        jb = (res.status & _ffi.ROW_JUMPBACK) != 0
        for s in range(len(lens)):
            m = (res.seq_index == s)
            j = np.nonzero(m & jb)[0]
            assert len(j) == len([x for x in range(0, lens[s] - inc + 1, inc) if x + w > lens[s]])
            if len(j):
                assert np.all(res.start[j] == lens[s] - w) and np.all(res.stop[j] == lens[s])
                vals = res.kld[j]                                   # identical duplicates (all NaN if the tail is dropped)
                assert np.all((vals == vals[0]) | (np.isnan(vals) & np.isnan(vals[0])))
            reg = np.nonzero(m & ~jb)[0]
            assert np.array_equal(res.start[reg], 1 + inc * np.arange(len(reg)))
        # islands exist: the score distribution has a tail (C1 is a single 50 kb scaffold: too short to hold one)
        if shape != "C1":
            assert res.kld[k].max() > 1.3 * np.median(res.kld[k])


def test_rccl_allreduce_path_single_rank():
    """The nccl(=RCCL) branch of Engine.profile_allreduce in a one-rank process group (the 8-GPU run is the driver's): the
    all-reduce runs IN PLACE on the library's raw-profile buffer (zero-copy tensor view) under the context's HIP stream
    (torch.cuda.ExternalStream); the copying path (export -> all_reduce -> import), which a job falls back to when
    ProcessGroupNCCL refuses the view or the first-use comparison fails, gives the same profile."""
    import torch.distributed as dist
    seqs = synth_seqs([40000, 7000], 13, n_frac=0.05)
    with make_engine(1, 8) as e:
        e.load(seqs)
        e.profile_reset(); e.profile_add()
        before = e.profile_raw()
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29611", rank=0, world_size=1)
        try:
            e.profile_allreduce(force=True)
            assert e.allreduce_path == "in_place_external_stream"
            assert np.array_equal(e.profile_raw(), before)
            e._allreduce_mode = "copy"                       # what a failed first-use check leaves behind
            e.profile_allreduce(force=True)
            assert e.allreduce_path == "export_import_copy"
        finally:
            dist.destroy_process_group()
        assert np.array_equal(e.profile_raw(), before)
        e.profile_finalize()
        from oracle import frisk_oracle_np as N
        sym, tl, ex, nn = e.profile_get()
        osym, ometa = N.genome_profile(seqs, 1, 8)
        assert np.array_equal(sym, osym) and (tl, ex, nn) == tuple(ometa)


def test_library_allreduce_over_a_raw_rccl_communicator():
    """frisk_profile_allreduce(ctx, ncclComm_t): the one collective without torch.distributed - a one-rank communicator made with
    ncclGetUniqueId / ncclCommInitRank through ctypes ("RCCL communicator is created by the python side and passed
    opaquely"), the all-reduce enqueued on the context's stream between profile_add and finalize.  NULL = single GPU, no-op."""
    import ctypes as C
    seqs = synth_seqs([30000, 9000], 14, n_frac=0.05, lower_frac=0.1)
    with make_engine(1, 8) as e:
        e.load(seqs)
        e.profile_reset(); e.profile_add()
        before = e.profile_raw()
        e.profile_allreduce(comm=0)                          # NULL communicator
        assert e.allreduce_path == "rccl_direct" and np.array_equal(e.profile_raw(), before)
        rccl = C.CDLL("librccl.so.1")

        class UniqueId(C.Structure):
            _fields_ = [("internal", C.c_char * 128)]
        uid, comm = UniqueId(), C.c_void_p()
        rccl.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
        rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
        assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
        try:
            e.profile_allreduce(comm=comm)
            e.profile_finalize()                             # enqueued behind the collective, no host wait in between
            assert np.array_equal(e.profile_raw(), before)
        finally:
            rccl.ncclCommDestroy(comm)
        from oracle import frisk_oracle_np as N
        sym, tl, ex, nn = e.profile_get()
        osym, ometa = N.genome_profile(seqs, 1, 8)
        assert np.array_equal(sym, osym) and (tl, ex, nn) == tuple(ometa)


def test_native_fasta_reader_matches_python_reader(tmp_path):
    """frisk_fasta_load (C++, zlib) against frisk_amd.fasta.iterFasta (the restated iterFasta, L139-164)."""
    import gzip
    import os
    from frisk_amd import _ffi
    from frisk_amd.fasta import readFasta
    from golden_util import INPUTS

    def canon(b):
        return bytes(c if c in b"ACGTacgt" else ord("N") for c in b)

    tricky = tmp_path / "tricky.fa"
    tricky.write_bytes(b"stray text before any header\nACGT\n>>r1>  first record   desc >\r\n  ACGTNN  \r\n\r\nac gt\n\n>r2\n>r3\tx\n"
                       b"TTTTTTTTTT\nGG>GG\n>r4 last no newline\nACGTRYKM")
    gz = tmp_path / "tricky.fa.gz"
    with gzip.open(gz, "wb") as fh:
        fh.write(tricky.read_bytes())
    big = tmp_path / "big.fa"
    with open(big, "wb") as fh:                 # > 64 records, long lines, crosses the 4 MiB read buffer
        for i in range(70):
            fh.write(b">c%d\n" % i + (b"ACGTTGCAAG" * 30000 if i == 3 else b"ACGTN" * (i + 1)) + b"\n")
    paths = [str(tricky), str(gz), str(big)] + [os.path.join(INPUTS, f) for f in ("kat.fa", "smalls.fa", "k8.fa")]
    with make_engine(1, 4) as e:
        for p in paths:
            names = e.load_fasta(p)
            pn, ps = readFasta(p)
            assert names == pn, p
            assert e.seq_lens == [len(s) for s in ps], p
            for i in (0, len(ps) // 2, len(ps) - 1):
                assert e.read_seq(i) == canon(ps[i]), (p, i)
        with pytest.raises(_ffi.FriskHipError):
            e.load_fasta(str(tmp_path / "missing.fa"))
        bad = tmp_path / "bad.fa"
        bad.write_bytes(b">\nACGT\n")
        with pytest.raises(_ffi.FriskHipError):
            e.load_fasta(str(bad))


def test_many_small_scaffolds():
    """Hundreds of scaffolds (host staging path of frisk_seq_load, descriptor search in the scan kernel, skipped and
    rescued small scaffolds in between)."""
    rng = np.random.default_rng(3)
    lens = [int(x) for x in rng.integers(200, 9000, size=300)]
    seqs = synth_seqs(lens, 31, island_frac=0.2, n_frac=0.1, lower_frac=0.05)
    for rescue in (False, True):
        (sym, meta), rows = oracle_rows(seqs, 1, 8, 2000, 500, scaffolds_all=rescue)
        with make_engine(1, 8) as e:
            e.load(seqs)
            e.profile_reset(); e.profile_add(); e.profile_finalize()
            gsym, tl, ex, nn = e.profile_get()
            assert np.array_equal(gsym, sym) and (tl, ex, nn) == tuple(meta)
            res = e.scan(2000, 500, scaffolds_all=rescue)
            kept = np.nonzero(res.kept)[0]
            assert len(kept) == len(rows) > 500
            got = [(str(res.seq_index[r]), int(res.start[r]), int(res.stop[r])) for r in kept.tolist()]
            assert got == [(r["name"], r["start"], r["stop"]) for r in rows]
            worst = max(abs(float(res.kld[r]) - exp["KLD"]) for r, exp in zip(kept.tolist(), rows))
            assert worst <= KLD_TOL


@pytest.mark.parametrize("shape", ["C1", "C2", "C3", "C4", "C5", "C5/8 unmasked repeats", "C5/8 soft-masked repeats",
                                   "C5/8 repeats of period 1-6 + satellite arrays"])
def test_full_size_rows_against_c_oracle(shape):
    """BASELINE configs at full size, row by row against the compiled CPU oracle (oracle/frisk_oracle_c.c, pinned to the
    reference's golden vectors by tests/test_oracle_c.py): profile bit-exact, kept set / coordinates / GC bit-exact,
    KLD to 1e-11.  C2..C4: every window.  C5 (3.3 Gb on one GPU): the full profile and three 20 000-candidate slices."""
    from oracle import frisk_oracle_c as OC
    from frisk_amd import _ffi, synth
    if shape == "C1":
        lens, kmin, kmax, w, inc, nfrac, slices = [50000], 1, 4, 5000, 1000, 0.0, [(0, -1)]
    elif shape == "C2":
        lens, kmin, kmax, w, inc, nfrac, slices = synth.C2_LENS, 1, 6, 5000, 500, 0.0, [(0, -1)]
    elif shape == "C3":
        lens, kmin, kmax, w, inc, nfrac, slices = synth.C3_LENS, 1, 8, 5000, 1000, 0.001, [(0, -1)]
    elif shape == "C4":
        lens, kmin, kmax, w, inc, nfrac, slices = synth.C4_LENS, 1, 8, 2000, 500, 0.07, [(0, -1)]
    elif shape == "C5":
        lens = [n for r in range(8) for n in synth.c5_shard_lens(8, r)]
        kmin, kmax, w, inc, nfrac = 1, 8, 5000, 1000, 0.07
        slices = [(0, 20000), (1_600_000, 1_620_000), (3_260_000, 3_280_000)]
    else:
        # the bench's "realistic" shapes on the bench's shard (410 Mb): simple repeats at a primate-like density - poly-A / T tails
        # and microsatellites that wrap 4-bit counters in almost half the windows of the unmasked form (the adaptive width then
        # takes 8-bit counters for the bulk), and a soft-masked form in which the reference's 30 % filter drops most windows
        lens = synth.c5_shard_lens(8, 0)
        kmin, kmax, w, inc, nfrac = 1, 8, 5000, 1000, 0.07
        slices = [(0, 30000), (200_000, 230_000), (380_000, 410_000)]
    sat_slices = 0
    with make_engine(kmin, kmax) as e:
        seed = {"C1": 1, "C2": 2, "C3": 3, "C4": 4, "C5": 5}.get(shape, 0xC5)
        kw = dict(island_frac=0.02, n_frac=nfrac, lower_frac=0.01)
        if shape == "C5/8 unmasked repeats":
            kw = dict(synth.REPEATS_UNMASKED)
        elif shape == "C5/8 soft-masked repeats":
            kw = dict(synth.REPEATS_SOFT)
        elif shape.startswith("C5/8 repeats of period"):    # the shape the side table was NOT designed around (synth.py)
            kw = dict(synth.REPEATS_MIXED)
        if shape in ("C1", "C2", "C3"):
            # the oracle's input is generated on the HOST (frisk_amd/synth.py) and uploaded as ASCII: the GPU's pack, its
            # counters and its scores are all on the checked path, nothing the oracle sees has passed through the device
            host = synth_seqs(lens, seed, **kw)
            e.load(host)
            S = OC.Seqs(host)
        else:
            e.synth(lens, seed=seed, **kw)
            S = None
        e.profile_reset(); e.profile_add(); e.profile_finalize()
        sym, tl, ex, nn = e.profile_get()
        res = e.scan(w, inc, rip=True)
        if S is None:                   # 249 Mb / 3.3 Gb: generated on the device, read back (the generators are tested equal)
            S = OC.Seqs([e.read_seq(q) for q in range(len(lens))])
        if shape.startswith("C5/8 repeats of period"):
            # ... and slices INSIDE satellite arrays of the largest scaffold (every window there wraps 4-bit counters with max-mers
            # of no short period: the hand-over chain's case), found from the generator's own specification
            sat, _ = synth._satellite(lens[0], seed, 0, kw["sat_frac"])
            edges = np.nonzero(np.diff(sat.astype(np.int8)) == 1)[0]
            assert edges.size >= 2 and abs(sat.mean() - kw["sat_frac"]) < 0.02
            for pos in edges[:3].tolist():
                a = max(0, pos // inc - 10)
                slices.append((a, a + 400))
                sat_slices += 1
        osym, ometa = OC.genome_profile(S, kmin, kmax)
        assert np.array_equal(sym, osym) and (tl, ex, nn) == tuple(ometa)
        ig = OC.genome_ivom(osym, ometa, kmin, kmax)
        checked = 0
        for a, b in slices:
            exp = OC.scan(S, ig, kmin, kmax, w, inc, rip=True, cand=(a, b))
            sl = slice(a, None if b < 0 else b)
            k = np.nonzero(res.kept[sl])[0] + a
            assert len(k) == len(exp["kld"])
            assert np.array_equal(res.seq_index[k], exp["seq"])
            assert np.array_equal(res.start[k], exp["start"]) and np.array_equal(res.stop[k], exp["stop"])
            assert np.array_equal(res.gc[k], exp["gc"])
            for col in ("pi", "si", "cri"):
                assert np.array_equal(getattr(res, col)[k], exp[col], equal_nan=True)
            assert not np.any(exp["status"] & OC.ROW_ZERO_DIV) and not np.any(res.zero_weight[k])
            assert np.array_equal((res.status[k] & _ffi.ROW_NO_MAXMER) != 0, (exp["status"] & OC.ROW_NO_MAXMER) != 0)
            assert np.max(np.abs(res.kld[k] - exp["kld"])) <= 1e-11
            checked += len(k)
        assert checked > (5000 if shape != "C1" else 40)
        if shape == "C5/8 unmasked repeats":        # the path real assemblies take: 4-bit counters + the side table for period-4 max-mers
            assert e.scan_stat()[0] == 4 and e.scan_side()
        if sat_slices:
            assert e.scan_stat()[1] > 3000      # windows handed from 4-bit to 8-bit counters (the arrays: ~3 % of 410 k)


@pytest.mark.parametrize("want_rip", [False, True])
@pytest.mark.parametrize("kmin,kmax,w,inc", [(1, 8, 40000, 15000), (1, 8, 65535, 30000), (3, 8, 30000, 29000),
                                             (2, 7, 65535, 20000), (1, 4, 60000, 7000), (2, 8, 9000, 4000),
                                             (4, 8, 20000, 9000), (1, 6, 12000, 5000), (5, 5, 10000, 3000)])
def test_long_windows_against_c_oracle(kmin, kmax, w, inc, want_rip):
    """Windows beyond the unrolled fast paths (generic kernel), up to the 65 535 bases that 16-bit LDS counters hold, with
    many invalid runs (hence a long orphan list at K = 8, which displaces the shared prefix tables in LDS)."""
    from oracle import frisk_oracle_c as OC
    rng = np.random.default_rng(w + kmax)
    seqs = []
    for n in (250000, 70000, 66000):
        s = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n, p=[0.35, 0.15, 0.2, 0.3])
        for a in rng.integers(0, n - 40, size=n // 60):           # a short invalid run every ~60 bases
            s[a:a + int(rng.integers(1, 4))] = ord("N")
        s[1000:3000] |= 0x20                                       # a soft-masked stretch
        seqs.append(s.tobytes())
    with make_engine(kmin, kmax) as e:
        e.load(seqs)
        e.profile_reset(); e.profile_add(); e.profile_finalize()
        sym, tl, ex, nn = e.profile_get()
        rip = want_rip and kmin <= 2
        if want_rip and not rip:
            pytest.skip("RIP needs kmin <= 2")
        res = e.scan(w, inc, rip=rip)
        osym, ometa = OC.genome_profile(seqs, kmin, kmax)
        assert np.array_equal(sym, osym) and (tl, ex, nn) == tuple(ometa)
        exp = OC.scan(seqs, OC.genome_ivom(osym, ometa, kmin, kmax), kmin, kmax, w, inc, rip=rip)
        k = np.nonzero(res.kept)[0]
        assert len(k) == len(exp["kld"]) and len(k) >= 3
        assert np.array_equal(res.start[k], exp["start"]) and np.array_equal(res.stop[k], exp["stop"])
        assert np.array_equal(res.gc[k], exp["gc"])
        assert np.array_equal(res.status[k] & 0xB, (exp["status"] & 0xA) | 1)       # KEPT | ZERO_WEIGHT | NO_MAXMER
        if rip:
            for col in ("pi", "si", "cri"):
                assert np.array_equal(getattr(res, col)[k], exp[col], equal_nan=True)
        assert np.max(np.abs(res.kld[k] - exp["kld"])) <= 1e-11


@pytest.mark.parametrize("kmin,kmax,w,inc,scaffolds_all,rip", [
    (1, 8, 65536, 30000, False, True), (1, 8, 100000, 60000, True, False), (2, 6, 250000, 100000, False, True),
    (3, 8, 70000, 69000, True, False), (1, 3, 131072, 131072, False, True), (8, 8, 90000, 45000, False, False)])
def test_windows_beyond_lds_counters_against_c_oracle(kmin, kmax, w, inc, scaffolds_all, rip):
    """Windows longer than 65 535 bases take the global-memory path (scan_big_kernel.h: 32-bit tables of all orders, the
    order-K table walked instead of the positions); same comparisons as everywhere, plus the count tables themselves."""
    from oracle import frisk_oracle_c as OC
    rng = np.random.default_rng(w + 7 * kmax + kmin)
    seqs = []
    for n in (700000, 2 * w + 12345, w + w // 3, 4000):
        s = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n, p=[0.3, 0.2, 0.15, 0.35])
        for a in rng.integers(0, max(1, n - 3000), size=12):
            s[a:a + int(rng.integers(1, 2500))] = ord("N")
        s[500:1500] |= 0x20
        s[2000:2600] = ord("A")                                    # big counts
        seqs.append(s.tobytes())
    with make_engine(kmin, kmax) as e:
        e.load(seqs)
        e.profile_reset(); e.profile_add(); e.profile_finalize()
        sym, tl, ex, nn = e.profile_get()
        res = e.scan(w, inc, rip=rip, scaffolds_all=scaffolds_all, debug=True)
        res2 = e.scan(w, inc, rip=rip, scaffolds_all=scaffolds_all)
        osym, ometa = OC.genome_profile(seqs, kmin, kmax)
        assert np.array_equal(sym, osym) and (tl, ex, nn) == tuple(ometa)
        exp = OC.scan(seqs, OC.genome_ivom(osym, ometa, kmin, kmax), kmin, kmax, w, inc, scaffolds_all=scaffolds_all,
                      rip=rip, debug=True)
    k = np.nonzero(res.kept)[0]
    assert len(k) == len(exp["kld"]) and len(k) >= 3
    assert np.array_equal(res.seq_index[k], exp["seq"])
    assert np.array_equal(res.start[k], exp["start"]) and np.array_equal(res.stop[k], exp["stop"])
    assert np.array_equal(res.counts[k].astype(np.int64), exp["counts"].astype(np.int64))
    assert np.array_equal(res.meta[k], exp["meta"])
    assert np.array_equal(res.gc[k], exp["gc"])
    assert np.array_equal(res.status[k] & 0xB, (exp["status"] & 0xA) | 1)
    if rip:
        for col in ("pi", "si", "cri"):
            assert np.array_equal(getattr(res, col)[k], exp[col], equal_nan=True)
    assert np.max(np.abs(res.kld[k] - exp["kld"])) <= 1e-11
    assert np.array_equal(res.kld, res2.kld, equal_nan=True) and np.array_equal(res.status, res2.status)


def _genome_with_repeats(n, every, units=(b"A", b"T", b"CA", b"TG", b"AAT", b"GATA", b"TTAGGG"), lens=(24, 40, 60, 90, 300, 700)):
    """One synthetic scaffold with microsatellites and poly-A runs sprinkled in, one about every `every` bases - a few of
    them longer than 8-bit counters hold."""
    from frisk_amd import synth
    s = np.frombuffer(synth.scaffold(n, 77, 0, island_frac=0.05, n_frac=0.01, lower_frac=0.01), dtype=np.uint8).copy()
    rng = np.random.default_rng(every)
    for a in range(5000, n - 3000, every):
        a += int(rng.integers(0, every // 3))
        u = units[int(rng.integers(0, len(units)))]
        ln = int(rng.choice(lens))                                             # 300+: wraps 8-bit counters as well
        rep = (u * (ln // len(u) + 1))[:ln]
        s[a:a + ln] = np.frombuffer(rep, dtype=np.uint8)
    return [s.tobytes()]


@pytest.mark.parametrize("every,units,lens,expect_bulk,expect_side", [
    (60000, None, None, 4, False), (9000, None, None, 4, True),
    (3000, (b"AAT", b"CAG", b"TTAGGG", b"TTCCG", b"AAAAT"), (60, 90, 150, 900), 8, False),
    (9000, (b"A", b"T", b"CA", b"TG", b"GATA", b"AAAT"), (24, 40, 60, 90, 150, 700), 4, True)])
def test_adaptive_counter_width_and_handover(every, units, lens, expect_bulk, expect_side):
    """The default K = 8 kernel counts max-mers in 4- or 8-bit counters (scan8_kernel.h) and hands windows with a more
    frequent max-mer to the next wider form: 4-bit -> 8-bit -> 16-bit.  A 12 Mb genome with microsatellites and poly-A
    runs (a few of them longer than 8-bit counters hold) sprinkled in: every row must still equal the compiled oracle's,
    the hand-over lists must have been used, and the sample must have picked the expected form for the bulk: plain 4-bit
    counters where repeats are sparse, 4-bit counters with the side table for max-mers of period <= 4 where simple repeats are
    dense, 8-bit counters where the dense repeats are of periods 3, 5 and 6, which the side table does not hold."""
    from oracle import frisk_oracle_c as OC
    from frisk_amd import _ffi
    # (14 Mb: a scan of 12 288+ windows goes in chunks, and the tables slide inside a chunk; 12 Mb: window by window)
    seqs = _genome_with_repeats(12_000_000, every) if units is None else _genome_with_repeats(14_000_000 if expect_side else 12_000_000, every, units, lens)
    with make_engine(1, 8) as e:
        e.load(seqs)
        e.profile_reset(); e.profile_add(); e.profile_finalize()
        sym, tl, ex, nn = e.profile_get()
        res = e.scan(5000, 1000, rip=True)
        stat = [int(_ffi.lib().frisk_last_scan_stat(e._ctx, i)) for i in range(3)]
        sided = e.scan_side()
        sub = e.scan(5000, 1000, rip=True, c0=1234, c1=4321)                   # another range: another sample, other widths
        few = e.scan(5000, 1000, rip=True, c0=700, c1=1000)                     # same batch and geometry: the first sample's choice holds
        stat_few = [int(_ffi.lib().frisk_last_scan_stat(e._ctx, i)) for i in range(3)]
        e.load(seqs)                                                            # a new residency forgets the choice ...
        e.profile_reset(); e.profile_add(); e.profile_finalize()
        other = e.scan(5000, 1000, rip=True, c0=2000, c1=2900)                  # ... and 900 windows are too few to sample: 8-bit bulk
        stat_other = [int(_ffi.lib().frisk_last_scan_stat(e._ctx, i)) for i in range(3)]
    assert stat[0] == expect_bulk and sided == expect_side, (stat, sided)
    assert stat[1] > 50 and stat[2] > 5, stat                                  # both hand-over lists were used
    assert stat_few[0] == expect_bulk, stat_few
    for f in ("start", "stop", "status", "kld", "gc", "pi", "si", "cri"):        # same bits whichever form scored the window
        assert np.array_equal(getattr(sub, f), getattr(res, f)[1234:4321], equal_nan=True), f
        assert np.array_equal(getattr(few, f), getattr(res, f)[700:1000], equal_nan=True), f
        assert np.array_equal(getattr(other, f), getattr(res, f)[2000:2900], equal_nan=True), f     # 8-bit bulk == 4-bit bulk, bit for bit
    assert stat_other[0] == 8 and stat_other[1] == 0, stat_other
    S = OC.Seqs(seqs)
    osym, ometa = OC.genome_profile(S, 1, 8)
    assert np.array_equal(sym, osym) and (tl, ex, nn) == tuple(ometa)
    exp = OC.scan(S, OC.genome_ivom(osym, ometa, 1, 8), 1, 8, 5000, 1000, rip=True)
    k = np.nonzero(res.kept)[0]
    assert len(k) == len(exp["kld"])
    assert np.array_equal(res.start[k], exp["start"]) and np.array_equal(res.stop[k], exp["stop"])
    assert np.array_equal(res.gc[k], exp["gc"])
    for col in ("pi", "si", "cri"):
        assert np.array_equal(getattr(res, col)[k], exp[col], equal_nan=True)
    assert np.max(np.abs(res.kld[k] - exp["kld"])) <= 1e-11


def test_long_scan_in_two_row_segments():
    """A scan of more than 2^17 candidates runs its last sixteenth as a second launch on a second stream while the rows of
    the first fifteen go to the host; chunks are dealt by per-XCD counters.  133 k windows over a genome with repeats that
    wrap 4- and 8-bit counters: both segments hand windows on, the rows equal those of four shorter scans (one segment
    each) bit for bit, and the rows on both sides of the cut equal the compiled oracle's."""
    from oracle import frisk_oracle_c as OC
    seqs = _genome_with_repeats(12_000_000, 60000)
    w, inc = 5000, 90
    with make_engine(1, 8) as e:
        e.load(seqs)
        e.profile_reset(); e.profile_add(); e.profile_finalize()
        sym, tl, ex, nn = e.profile_get()
        res = e.scan(w, inc)
        stat = e.scan_stat()
        n = res.n_candidates
        assert n > (1 << 17) and stat[3] == 2 and stat[1] > 100 and stat[2] > 10, (n, stat)
        cuts = [0, n // 4, n // 2, n - n // 4, n]
        for a, b in zip(cuts, cuts[1:]):
            part = e.scan(w, inc, c0=a, c1=b)
            assert e.scan_stat()[3] == 1
            for f in ("seq_index", "start", "stop", "status", "kld", "gc"):
                assert np.array_equal(getattr(part, f), getattr(res, f)[a:b], equal_nan=True), (f, a, b)
    S = OC.Seqs(seqs)
    osym, ometa = OC.genome_profile(S, 1, 8)
    ig = OC.genome_ivom(osym, ometa, 1, 8)
    unit = 16 * 16                                                          # (2 i <= w - 7: the tables slide, chunks of 16)
    cut = (n // unit - max(1, n // unit // 16)) * unit                      # where frisk_scan cuts (whole chunks, stride 16)
    for a, b in ((cut - 1500, cut + 1500), (n - 2000, n), (0, 2000)):
        exp = OC.scan(S, ig, 1, 8, w, inc, cand=(a, b))
        k = a + np.nonzero(res.kept[a:b])[0]
        assert len(k) == len(exp["kld"])
        assert np.array_equal(res.start[k], exp["start"]) and np.array_equal(res.stop[k], exp["stop"])
        assert np.array_equal(res.gc[k], exp["gc"])
        assert np.max(np.abs(res.kld[k] - exp["kld"])) <= 1e-11


@pytest.mark.parametrize("w,inc,scaffolds_all,kmin,kmax", [(5000, 1000, False, 1, 8), (2000, 2500, True, 1, 8), (3000, 700, True, 2, 6),
                                                          (5000, 1000, True, 1, 8)])
def test_window_tile_sharding_equals_one_gpu(tmp_path, w, inc, scaffolds_all, kmin, kmax):
    """frisk_fasta_load_shard (window tiles + halo): the ranks of a 2-, 3- and 7-GPU job are played one after
    the other on this GPU.  The sum of their raw profiles must be the one-GPU profile bit for bit (every base owned once),
    and their rows, concatenated in rank order, must be the one-GPU rows bit for bit - with only the tiles resident."""
    from frisk_amd import synth
    lens = [333_337, 40_000, 7_001, 5_000, 0, 1_234, 120_500, 9_999, 64_000]
    seqs = synth_seqs(lens, 41, island_frac=0.1, n_frac=0.08, lower_frac=0.05)
    fa = tmp_path / "g.fa"
    with open(fa, "wb") as fh:
        for i, s in enumerate(seqs):
            fh.write(b">s%d some description\n" % i)
            for o in range(0, len(s), 70):
                fh.write(s[o:o + 70] + b"\n")
    rip = kmin <= 2
    with make_engine(kmin, kmax) as e:
        names = e.load_fasta(str(fa))
        e.profile_reset(); e.profile_add(); whole_raw = e.profile_raw(); e.profile_finalize()
        full = e.scan(w, inc, rip=rip, scaffolds_all=scaffolds_all)
        whole_bases = e.padded_len
        for world in (2, 3, 7):
            raws, parts, ranges, resident = [], [], [], []
            for rank in range(world):
                nm, (c0, c1) = e.load_fasta_shard(str(fa), w, inc, rank, world, scaffolds_all)
                assert nm == names and e.seq_lens == lens
                resident.append(e.padded_len)
                e.profile_reset(); e.profile_add()
                raws.append(e.profile_raw())
                ranges.append((c0, c1))
            assert np.array_equal(np.sum(raws, axis=0), whole_raw)
            assert ranges[0][0] == 0 and ranges[-1][1] == full.n_candidates
            assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
            assert max(resident) < 0.75 * whole_bases                  # a rank holds its share, not the genome
            for rank in range(world):
                e.load_fasta_shard(str(fa), w, inc, rank, world, scaffolds_all)
                e.profile_set_raw(whole_raw); e.profile_finalize()     # what the all-reduce leaves on every rank
                r = e.scan(w, inc, rip=rip, scaffolds_all=scaffolds_all)
                assert len(r) == ranges[rank][1] - ranges[rank][0]
                parts.append(r)
            for f in ("seq_index", "start", "stop", "status", "kld", "gc") + (("pi", "si", "cri") if rip else ()):
                cat = np.concatenate([getattr(p, f) for p in parts])
                assert np.array_equal(cat, getattr(full, f), equal_nan=True), (world, f)
        with pytest.raises(Exception):
            e.scan(w + 1, inc)                                          # the tiles were cut for another geometry


def test_staged_residency_and_packed_form():
    """frisk_seq_stage / _stage_packed / _commit / _export_packed: a batch uploaded into the second slot while another is
    resident and being scanned gives, once committed, exactly the results of a plain load - from ASCII (page-locked and
    pageable sources) and from the 0.5 B/base packed arrays; the resident batch is untouched until the commit."""
    seqs_a = synth_seqs([90_000, 12_345, 0, 7_000], 51, island_frac=0.2, n_frac=0.1, lower_frac=0.1)
    seqs_b = synth_seqs([64_000, 33_333, 5_001], 52, island_frac=0.1, n_frac=0.05, lower_frac=0.2)

    def run(e):
        e.profile_reset(); e.profile_add(); e.profile_finalize()
        return e.profile_raw(), e.scan(5000, 1000, rip=True)

    def same(x, y):
        assert np.array_equal(x[0], y[0])
        for f in ("seq_index", "start", "stop", "status", "kld", "gc", "pi", "si", "cri"):
            assert np.array_equal(getattr(x[1], f), getattr(y[1], f), equal_nan=True), f

    with make_engine(1, 8) as e:
        e.load(seqs_a); ref_a = run(e)
        e.load(seqs_b); ref_b = run(e)
        # pageable sources: B resident, A staged; B's results must not move while A is in flight
        e.stage(seqs_a)
        same(run(e), ref_b)
        e.commit()
        assert e.seq_lens == [len(s) for s in seqs_a]
        assert e.read_seq(1) == bytes(c if c in b"ACGTacgt" else ord("N") for c in seqs_a[1])
        same(run(e), ref_a)
        # page-locked sources (asynchronous copies), several rounds of ping-pong
        pinned = []
        for i, s in enumerate(seqs_b):
            a = e.host_array("b%d" % i, max(len(s), 1))
            a[:len(s)] = np.frombuffer(s, dtype=np.uint8)
            pinned.append(a[:len(s)])
        for _ in range(3):
            e.stage(pinned); same(run(e), ref_a); e.commit(); same(run(e), ref_b)
            e.stage(seqs_a); e.commit(); same(run(e), ref_a)
        # the packed form: export B's arrays, stage them over A, commit with names
        e.stage(pinned); e.commit()
        codes, inv, low = e.export_packed()
        assert codes.size == 2 * (e.padded_len // 32) and inv.size == low.size == e.padded_len // 32
        e.load(seqs_a)
        e.stage_packed(codes, inv, low, [len(s) for s in seqs_b])
        same(run(e), ref_a)
        e.commit(names=["x", "y", "z"])
        same(run(e), ref_b)
        assert e._lib.frisk_seq_name(e._ctx, 1) == b"y"
        with pytest.raises(Exception):
            e.commit()                                  # nothing staged


@pytest.mark.parametrize("kmin,kmax,w,inc,rip", [(1, 9, 3000, 1000, True), (2, 10, 5000, 2500, True), (9, 9, 2000, 1000, False),
                                                 (7, 11, 4000, 4000, False)])
def test_orders_above_eight_against_c_oracle(kmin, kmax, w, inc, rip):
    """-k above 8 (the reference's -k is unbounded, L1197-1206): the global-memory paths - profile with one global atomic
    per position, window tables of all orders in a scratch slice per workgroup - against the compiled oracle: profile
    bit-exact, rows bit-exact on integers / GC / RIP, KLD to 1e-11, count tables of a few windows."""
    from oracle import frisk_oracle_c as OC
    seqs = synth_seqs([60_000, 9_000, 20_011, 300], 61, island_frac=0.2, n_frac=0.06, lower_frac=0.05)
    with make_engine(kmin, kmax) as e:
        e.load(seqs)
        e.profile_reset(); e.profile_add(); e.profile_finalize()
        sym, tl, ex, nn = e.profile_get()
        res = e.scan(w, inc, rip=rip and kmin <= 2)
        dbg = e.scan(w, inc, c0=3, c1=6, debug=True)
        half = e.padded_len // 2 // 32 * 32
        e.profile_reset(); e.profile_add(pos_begin=0, pos_end=half); e.profile_add(pos_begin=half, pos_end=e.padded_len)
        e.profile_finalize()
        sym2 = e.profile_get()[0]
    osym, ometa = OC.genome_profile(seqs, kmin, kmax)
    assert np.array_equal(sym, osym) and (tl, ex, nn) == tuple(ometa)
    assert np.array_equal(sym2, osym)                                           # linear over position ranges here too
    ig = OC.genome_ivom(osym, ometa, kmin, kmax)
    exp = OC.scan(seqs, ig, kmin, kmax, w, inc, rip=rip and kmin <= 2)
    exp_dbg = OC.scan(seqs, ig, kmin, kmax, w, inc, cand=(3, 6), debug=True)           # (the count tables are 4^K wide: three rows)
    k = np.nonzero(res.kept)[0]
    assert len(k) == len(exp["kld"]) and len(k) >= 8
    assert np.array_equal(res.start[k], exp["start"]) and np.array_equal(res.stop[k], exp["stop"])
    assert np.array_equal(res.gc[k], exp["gc"])
    if rip and kmin <= 2:
        for col in ("pi", "si", "cri"):
            assert np.array_equal(getattr(res, col)[k], exp[col], equal_nan=True)
    assert np.max(np.abs(res.kld[k] - exp["kld"])) <= 1e-11
    kd = np.nonzero(dbg.kept)[0]
    assert len(kd) == len(exp_dbg["kld"]) >= 1
    assert np.array_equal(dbg.counts[kd].astype(np.int64), exp_dbg["counts"].astype(np.int64))
    assert np.array_equal(dbg.meta[kd], exp_dbg["meta"])


@pytest.mark.parametrize("kmin,kmax,w,inc,rip,n", [(1, 8, 5000, 1000, True, 3_000_000), (1, 8, 3000, 20, False, 3_000_000),
                                                    (2, 6, 2000, 500, True, 1_000_000), (1, 4, 5000, 1000, True, 400_000)])
def test_page_locked_result_buffers_equal_ordinary_ones(kmin, kmax, w, inc, rip, n):
    """Engine.scan(pinned=True) returns views of page-locked buffers (frisk_host_alloc: what bench.py times); the default is
    ordinary numpy arrays.  Same bits either way - short and long (two row segments) scans, and the 16-bit form (k <= 4)."""
    from frisk_amd import synth
    seqs = [synth.scaffold(n, 5, 0, island_frac=0.05, n_frac=0.03, lower_frac=0.02), synth.scaffold(7000, 6, 0)]
    with make_engine(kmin, kmax) as e:
        e.load(seqs)
        e.profile_reset(); e.profile_add(); e.profile_finalize()
        a = e.scan(w, inc, rip=rip, scaffolds_all=True)
        b = e.scan(w, inc, rip=rip, scaffolds_all=True, pinned=True)
        cols = ("seq_index", "start", "stop", "status", "kld", "gc") + (("pi", "si", "cri") if rip else ())
        for f in cols:
            assert np.array_equal(getattr(a, f), np.array(getattr(b, f)), equal_nan=True), f
        assert a.n_candidates == b.n_candidates and int(a.kept.sum()) > 0

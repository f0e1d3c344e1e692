"""The 0.25 B/base upload form on the GPU (frisk_pack_2bit -> frisk_seq_stage_2bit -> frisk_seq_commit): a batch
staged as 2-bit codes + run lists, its codes crossing PCIe in pieces with phase A following the pieces, gives the SAME three
resident arrays as frisk_seq_load's device packer, the same profile and the same rows - whatever the piece size, for every
byte value, IUPAC letters, lowercase n, empty scaffolds and runs that cross word and piece boundaries."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def synth_seqs(lens, seed, **kw):
    from frisk_amd import synth
    return [synth.scaffold(n, seed, i, **kw) for i, n in enumerate(lens)]


def resident(e):
    return [a.copy() for a in e.export_packed()]


def run(e, piecewise=True):
    e.profile_reset()
    e.profile_add()
    e.profile_finalize()
    return e.profile_raw(), e.scan(5000, 1000, rip=True)


def same(x, y):
    assert np.array_equal(x[0], y[0]), "raw profiles differ"
    for f in ("seq_index", "start", "stop", "status", "kld", "gc", "pi", "si", "cri"):
        assert np.array_equal(getattr(x[1], f), getattr(y[1], f), equal_nan=True), f


def awkward_batch():
    rng = np.random.default_rng(11)
    allb = bytes(range(1, 256))
    s = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 40_000)
    for a in range(0, 40_000 - 1024, 1024):               # runs over every 32-position word boundary and every 1 024-base piece cut
        s[max(0, a - 17):a + 19] = ord("N")
        s[a + 100:a + 100 + 37] |= 0x20
        s[a + 500] = ord("n")
        s[a + 510:a + 515] = np.frombuffer(b"RYKMn", dtype=np.uint8)
    return [allb * 3, b"", s.tobytes(), b"A", b"", b"n" * 33 + b"acgt" * 20 + b"N" * 95, b"ACGTacgtNnRYKMSWBDHV-*"]


@pytest.mark.parametrize("piece_bases", [0, 32, 64, 1024, 4096 + 32, 1 << 20])
def test_staged_2bit_batch_is_the_loaded_batch(piece_bases):
    from frisk_amd import Engine
    seqs = awkward_batch()
    with Engine(1, 8) as e:
        e.load(seqs)
        want = resident(e)
        e.load([b"ACGT" * 50])                              # something else resident while the 2-bit form is staged
        codes, ri, rl, lens = e.pack_2bit(seqs)
        assert lens == [len(s) for s in seqs]
        e.stage_2bit(codes, ri, rl, lens, piece_bases=piece_bases)
        e.commit()
        got = resident(e)                                   # (waits for the last piece)
        for a, b, name in zip(want, got, ("codes", "inv", "low")):
            assert np.array_equal(a, b), name
        for i, s in enumerate(seqs):
            assert e.read_seq(i) == bytes(c if c in b"ACGTacgt" else ord("N") for c in s)
        # ... and back out in the same form
        c2, ri2, rl2 = e.export_2bit()
        assert np.array_equal(c2, codes) and np.array_equal(ri2, ri) and np.array_equal(rl2, rl)


@pytest.mark.parametrize("piece_bases", [0, 4096, 50_000])
@pytest.mark.parametrize("kmax", [8, 6, 10])
def test_profile_follows_the_pieces_and_rows_match(piece_bases, kmax):
    """profile_add directly behind the commit (the streamed case: one kernel per piece) == profile of the loaded batch, and the
    scan behind it == the scan of the loaded batch; a second profile_add on the now settled batch is the plain one."""
    from frisk_amd import Engine
    seqs = synth_seqs([180_000, 12_345, 0, 7_000, 64_000], 61, island_frac=0.2, n_frac=0.08, lower_frac=0.15, repeats_per_kb=0.3)
    with Engine(1, kmax) as e:
        e.load(seqs)
        ref = run(e)
        codes, ri, rl, lens = e.pack_2bit(seqs)
        if kmax == 8 and piece_bases == 4096:                # ... and the rows are the ORACLE's (numpy restatement, pinned to the reference's goldens)
            from oracle import frisk_oracle_np as N
            prof = N.genome_profile(seqs, 1, 8)
            want = N.scan([(str(i), s) for i, s in enumerate(seqs)], prof, 1, 8, 5000, 1000, False, True)
            e.stage_2bit(codes, ri, rl, lens, piece_bases=piece_bases)
            e.commit()
            raw, res = run(e)
            k = np.nonzero(res.kept)[0]
            assert len(k) == len(want) > 200
            assert [(int(res.start[r]), int(res.stop[r])) for r in k.tolist()] == [(w["start"], w["stop"]) for w in want]
            assert max(abs(float(res.kld[r]) - w["KLD"]) for r, w in zip(k.tolist(), want)) <= 1e-10
            assert all(float(res.gc[r]) == w["GC"] for r, w in zip(k.tolist(), want))
        for _ in range(2):                                   # both batch slots
            e.stage_2bit(codes, ri, rl, lens, piece_bases=piece_bases)
            same(run(e), ref)                                # the resident batch is untouched while the upload is in flight
            e.commit()
            same(run(e), ref)                                # streamed: phase A piece by piece
            same(run(e), ref)                                # settled: one launch
        # mask_host and explicit ranges on a freshly committed batch (not the piecewise path: they wait for the upload)
        e.stage_2bit(codes, ri, rl, lens, piece_bases=piece_bases)
        e.commit()
        e.profile_reset(); e.profile_add(mask_host=True); a = e.profile_raw()
        e.load(seqs)
        e.profile_reset(); e.profile_add(mask_host=True); b = e.profile_raw()
        assert np.array_equal(a, b)
        e.stage_2bit(codes, ri, rl, lens, piece_bases=piece_bases)
        e.commit()
        P = e.padded_len
        e.profile_reset(); e.profile_add(pos_begin=0, pos_end=P // 64 * 32); e.profile_add(pos_begin=P // 64 * 32, pos_end=P)
        e.profile_finalize()
        assert np.array_equal(e.profile_raw(), ref[0])


def test_dense_masks_and_bad_runs():
    from frisk_amd import Engine, _ffi
    seqs = synth_seqs([50_000, 3_000], 62, island_frac=0.1, n_frac=0.1, lower_frac=0.4)
    with Engine(1, 8) as e:
        e.load(seqs)
        ref = run(e)
        want = resident(e)
        codes, ri, rl, lens = e.pack_2bit(seqs)
        # the dense form of a mask: the bitmap without its PAD bits
        P = e.padded_len
        pad = np.zeros(P, bool)
        off = 0
        for n in lens:
            pad[off + n] = True
            off += n + 1
        pad[off:] = True
        padw = np.packbits(pad).view(">u4").astype(np.uint32)
        low_dense = want[2] & ~padw
        inv_dense = want[1] & ~padw
        e.stage_2bit(codes, ri, low_dense, lens)
        e.commit()
        for a, b in zip(want, resident(e)):
            assert np.array_equal(a, b)
        e.stage_2bit(codes, inv_dense, low_dense, lens, piece_bases=2048)
        e.commit()
        same(run(e), ref)
        with pytest.raises(_ffi.FriskHipError):
            e.stage_2bit(codes, np.array([[5, 3]]), rl, lens)
        with pytest.raises(_ffi.FriskHipError):
            e.stage_2bit(codes, np.array([[0, P + 1]]), rl, lens)
        same(run(e), ref)                                    # a refused stage leaves the resident batch alone


def test_one_pass_profile_counter_overflow_falls_back():
    """Phase A at K = 8 over a long range (2^30+ positions; here forced: one_pass=True) counts in 16-bit LDS fields, one pass
    (profile_add16_kernel); 65 536+ copies of one 8-mer inside one workgroup's chunk wrap a field - found by the table's grand total, and the workgroup counts its chunk again in the two-half
    32-bit form.  A scaffold with a 200 kb poly-A stretch, a 150 kb (CA)n and a 100 kb (AAT)n between random sequence: the
    profile must be the oracle's to the last count, with and without --maskHost, whole and in position ranges."""
    from oracle import frisk_oracle_np as N
    from frisk_amd import Engine
    rng = np.random.default_rng(21)
    rnd = lambda n: rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), n).tobytes()      # noqa: E731
    s0 = rnd(50_000) + b"A" * 200_000 + rnd(30_000) + b"CA" * 75_000 + b"N" * 50 + b"aat" * 33_000 + rnd(70_000)
    s1 = rnd(40_000) + b"T" * 140_000
    seqs = [s0, s1]
    with Engine(1, 8) as e:
        e.load(seqs)
        for mask_host in (False, True):
            osym, ometa = N.genome_profile(seqs, 1, 8, mask_host=mask_host)
            for one_pass in (True, False):
                e.profile_reset(); e.profile_add(mask_host=mask_host, one_pass=one_pass); e.profile_finalize()
                sym, tl, ex, nn = e.profile_get()
                assert np.array_equal(sym, osym) and (tl, ex, nn) == tuple(ometa), (mask_host, one_pass)
        P = e.padded_len
        e.profile_reset()
        for a, b in ((0, 100_032), (100_032, 400_000), (400_000, P)):
            e.profile_add(pos_begin=a, pos_end=b, one_pass=True)
        e.profile_finalize()
        osym, ometa = N.genome_profile(seqs, 1, 8)
        assert np.array_equal(e.profile_get()[0], osym)

"""Randomised differential test of the HIP path against the compiled CPU oracle: geometry, word sizes, flags and
sequence make-up (N runs, soft-masked runs, IUPAC letters, tiny scaffolds) are drawn from a fixed seed; every output
column is compared (integers and GC / PI / SI / CRI bit-exact, KLD to 1e-11)."""
import numpy as np
import pytest

from frisk_amd import _ffi
from frisk_amd.engine import Engine
from oracle import frisk_oracle_c as OC

pytestmark = pytest.mark.gpu


def _random_case(rng, kmax_hi=8):
    kmax = int(rng.integers(1, kmax_hi + 1))
    kmin = int(rng.integers(1, kmax + 1))
    w = int(rng.choice([37, 64, 100, 333, 512, 1000, 2048, 2049, 5000, 5121, 8192, 8193, 12000, 66000]))
    inc = max(1, int(w * rng.choice([0.05, 0.2, 0.5, 0.9, 1.0, 1.6])))
    seqs = []
    for _ in range(int(rng.integers(1, 6))):
        n = int(rng.choice([0, 5, w // 2, w, w + 1, 2 * w + 3, 3 * w, 7 * w + int(rng.integers(0, w))]))
        p = rng.dirichlet([2, 2, 2, 2])
        s = rng.choice(np.frombuffer(b"ATGC", dtype=np.uint8), size=n, p=p)
        for _ in range(int(rng.integers(0, 6))):
            if n == 0:
                break
            a = int(rng.integers(0, n))
            ln = int(rng.choice([1, 2, 7, 8, 9, 40, w // 3 + 1]))
            kind = rng.integers(0, 4)
            if kind == 0:
                s[a:a + ln] = ord("N")
            elif kind == 1:
                s[a:a + ln] |= 0x20                           # soft-masked
            elif kind == 2:
                s[a:a + ln] = rng.choice(np.frombuffer(b"RYKMnrx-*", dtype=np.uint8), size=len(s[a:a + ln]))
            else:
                s[a:a + ln] = s[a] if a < n else ord("A")      # a low-complexity run (big counts)
        seqs.append(s.tobytes())
    return dict(kmin=kmin, kmax=kmax, w=w, inc=inc, seqs=seqs, mask_host=bool(rng.integers(0, 2)),
                scaffolds_all=bool(rng.integers(0, 2)), rip=bool(rng.integers(0, 2)) and kmin <= 2 <= kmax)


@pytest.mark.parametrize("block", range(16))
def test_random_cases_against_c_oracle(block):
    rng = np.random.default_rng(1000 + block)
    checked = 0
    for case_no in range(25):
        c = _random_case(rng)
        tag = "block %d case %d: k=%d..%d w=%d i=%d mask=%s all=%s rip=%s lens=%s" % (
            block, case_no, c["kmin"], c["kmax"], c["w"], c["inc"], c["mask_host"], c["scaffolds_all"], c["rip"],
            [len(s) for s in c["seqs"]])
        with Engine(c["kmin"], c["kmax"]) as e:
            e.load(c["seqs"])
            e.profile_reset(); e.profile_add(mask_host=c["mask_host"]); e.profile_finalize()
            sym, tl, ex, nn = e.profile_get()
            osym, ometa = OC.genome_profile(c["seqs"], c["kmin"], c["kmax"], c["mask_host"])
            assert np.array_equal(sym, osym) and (tl, ex, nn) == tuple(ometa), tag
            res = e.scan(c["w"], c["inc"], rip=c["rip"], scaffolds_all=c["scaffolds_all"])
            ig = OC.genome_ivom(osym, ometa, c["kmin"], c["kmax"])
            exp = OC.scan(c["seqs"], ig, c["kmin"], c["kmax"], c["w"], c["inc"], scaffolds_all=c["scaffolds_all"], rip=c["rip"])
        k = np.nonzero(res.kept)[0]
        assert len(k) == len(exp["kld"]), tag
        if not len(k):
            continue
        assert np.array_equal(res.seq_index[k], exp["seq"]), tag
        assert np.array_equal(res.start[k], exp["start"]) and np.array_equal(res.stop[k], exp["stop"]), tag
        assert np.array_equal(res.gc[k], exp["gc"], equal_nan=True), tag
        zero = (exp["status"] & OC.ROW_ZERO_DIV) != 0
        assert np.array_equal((res.status[k] & _ffi.ROW_ZERO_WEIGHT) != 0, zero), tag
        assert np.array_equal((res.status[k] & _ffi.ROW_NO_MAXMER) != 0, (exp["status"] & OC.ROW_NO_MAXMER) != 0), tag
        if c["rip"]:
            for col in ("pi", "si", "cri"):
                assert np.array_equal(getattr(res, col)[k], exp[col], equal_nan=True), tag
        ok = ~zero
        if ok.any():
            assert np.max(np.abs(res.kld[k][ok] - exp["kld"][ok])) <= 1e-11, tag
        checked += len(k)
    assert checked > 100


@pytest.mark.parametrize("block", range(2))
def test_random_cases_up_to_k10(block):
    """The same differential test with the highest order drawn up to 10 (K = 9, 10: the global-memory kernels), on smaller
    inputs - those kernels walk 4^K bins per window."""
    rng = np.random.default_rng(9100 + block)
    checked = big = 0
    for case_no in range(12):
        c = _random_case(rng, kmax_hi=10)
        if c["kmax"] >= 9:
            c["w"] = min(c["w"], 5121)
            c["inc"] = max(1, min(c["inc"], c["w"]))
            c["seqs"] = [s[:4 * c["w"]] for s in c["seqs"][:3]]
            big += 1
        tag = "block %d case %d: k=%d..%d w=%d i=%d lens=%s" % (block, case_no, c["kmin"], c["kmax"], c["w"], c["inc"], [len(s) for s in c["seqs"]])
        with Engine(c["kmin"], c["kmax"]) as e:
            e.load(c["seqs"])
            e.profile_reset(); e.profile_add(mask_host=c["mask_host"]); e.profile_finalize()
            sym, tl, ex, nn = e.profile_get()
            osym, ometa = OC.genome_profile(c["seqs"], c["kmin"], c["kmax"], c["mask_host"])
            assert np.array_equal(sym, osym) and (tl, ex, nn) == tuple(ometa), tag
            res = e.scan(c["w"], c["inc"], rip=c["rip"], scaffolds_all=c["scaffolds_all"])
            ig = OC.genome_ivom(osym, ometa, c["kmin"], c["kmax"])
            exp = OC.scan(c["seqs"], ig, c["kmin"], c["kmax"], c["w"], c["inc"], scaffolds_all=c["scaffolds_all"], rip=c["rip"])
        k = np.nonzero(res.kept)[0]
        assert len(k) == len(exp["kld"]), tag
        if not len(k):
            continue
        assert np.array_equal(res.start[k], exp["start"]) and np.array_equal(res.stop[k], exp["stop"]), tag
        assert np.array_equal(res.gc[k], exp["gc"], equal_nan=True), tag
        zero = (exp["status"] & OC.ROW_ZERO_DIV) != 0
        assert np.array_equal((res.status[k] & _ffi.ROW_ZERO_WEIGHT) != 0, zero), tag
        ok = ~zero
        if ok.any():
            assert np.max(np.abs(res.kld[k][ok] - exp["kld"][ok])) <= 1e-11, tag
        checked += len(k)
    assert checked > 30 and big >= 1


@pytest.mark.parametrize("block", range(3))
def test_random_tile_sharding_equals_one_gpu(tmp_path, block):
    """frisk_fasta_load_shard with random geometry, word sizes, scaffold make-up and world size: the ranks of the job played one
    after the other give the one-GPU raw profile (summed) and the one-GPU rows (concatenated) bit for bit.  Every other rank
    reads its tiles through the seek index (frisk_fasta_load_shard_indexed) instead of parsing the file: the same resident
    words, the same profile share, the same rows."""
    from frisk_amd.fasta import writeFastaIndex
    rng = np.random.default_rng(4400 + block)
    for case_no in range(4):
        kmax = int(rng.choice([4, 6, 7, 8, 8]))
        kmin = int(rng.integers(1, max(2, kmax - 2)))
        w = int(rng.choice([400, 1000, 2000, 5000]))
        inc = max(1, int(w * rng.choice([0.1, 0.2, 0.5, 1.0, 1.3])))
        scaffolds_all = bool(rng.integers(0, 2))
        world = int(rng.choice([2, 3, 5, 8]))
        fa = tmp_path / ("g%d.fa" % case_no)
        with open(fa, "wb") as fh:
            for i in range(int(rng.integers(1, 7))):
                n = int(rng.choice([0, 7, w // 2, w + 1, 3 * w + 5, 11 * w + int(rng.integers(0, w)), 40 * inc + w]))
                s = rng.choice(np.frombuffer(b"ATGC", dtype=np.uint8), size=n, p=rng.dirichlet([2, 2, 2, 2]))
                for _ in range(int(rng.integers(0, 4))):
                    if n:
                        a = int(rng.integers(0, n))
                        ln = int(rng.choice([1, 9, 40, w // 3 + 1]))
                        if rng.integers(0, 2):
                            s[a:a + ln] = ord("N")
                        else:
                            s[a:a + ln] |= 0x20
                fh.write(b">scaf%d some words\n" % i)
                eol = b"\r\n" if (case_no + i) % 3 == 0 else b"\n"
                for o in range(0, n, 61):
                    fh.write(s[o:o + 61].tobytes() + eol)
                if i % 2:
                    fh.write(b"\n")                 # (a blank line between records)
        idx = tmp_path / ("g%d.fa.frisk.fai" % case_no)
        assert writeFastaIndex(str(fa), str(idx)) is not None
        tag = "block %d case %d: k=%d..%d w=%d i=%d all=%s world=%d" % (block, case_no, kmin, kmax, w, inc, scaffolds_all, world)
        rip = kmin <= 2 <= kmax
        with Engine(kmin, kmax) as e:
            e.load_fasta(str(fa))
            e.profile_reset(); e.profile_add(); whole_raw = e.profile_raw(); e.profile_finalize()
            full = e.scan(w, inc, rip=rip, scaffolds_all=scaffolds_all)
            raws, parts = [], []
            for rank in range(world):
                names_p, (c0, c1) = e.load_fasta_shard(str(fa), w, inc, rank, world, scaffolds_all, index=str(idx) if rank & 1 else None)
                assert (e.shard_index is not None) == bool(rank & 1), tag
                e.profile_reset(); e.profile_add()
                raws.append(e.profile_raw())
                if rank in (0, world - 1):          # both loaders leave the same words on the device
                    packed = [x.copy() for x in e.export_packed()]
                    names_i, cc = e.load_fasta_shard(str(fa), w, inc, rank, world, scaffolds_all, index=None if rank & 1 else [str(idx)])
                    assert names_i == names_p and cc == (c0, c1) and (e.shard_index is None) == bool(rank & 1), tag
                    assert all(np.array_equal(x, y) for x, y in zip(packed, e.export_packed())), tag
            assert np.array_equal(np.sum(raws, axis=0), whole_raw), tag
            for rank in range(world):
                e.load_fasta_shard(str(fa), w, inc, rank, world, scaffolds_all, index=None if rank & 1 else str(idx))
                e.profile_set_raw(whole_raw); e.profile_finalize()
                parts.append(e.scan(w, inc, rip=rip, scaffolds_all=scaffolds_all, chunks=bool(rank & 1)))
            for f in ("seq_index", "start", "stop", "status", "kld", "gc") + (("pi", "si", "cri") if rip else ()):
                cat = np.concatenate([getattr(p, f) for p in parts])
                assert np.array_equal(cat, getattr(full, f), equal_nan=True), (tag, f)

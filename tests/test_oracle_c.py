"""Pins the C oracle (oracle/frisk_oracle_c.c) to the golden vectors of the reference's own functions
(integer tables and coordinates bit-exact, KLD to 1e-12: the C sum runs over sorted max-mers, CPython's in
dict order) and to the numpy oracle on random inputs."""
import math

import numpy as np
import pytest

from oracle import frisk_oracle_c as OC
from golden_util import Case, case_names, same_float
from oracle import frisk_oracle as O
from oracle import frisk_oracle_np as N


@pytest.mark.parametrize("name", case_names())
def test_c_oracle_matches_reference_golden(name):
    c = Case(name)
    host = list(O.iter_fasta(c.host))
    sym, meta = OC.genome_profile([s for _, s in host], c.m, c.k, mask_host=c.mask_host)
    assert np.array_equal(sym, c.genome_counts)
    assert list(meta) == c.genome_meta
    ig = OC.genome_ivom(sym, meta, c.m, c.k)
    ig_np = N.genome_ivom_table(sym, meta, c.m, c.k)
    assert np.array_equal(ig, ig_np, equal_nan=True)
    query = list(O.iter_fasta(c.query)) if c.query else host
    got = OC.scan([s for _, s in query], ig, c.m, c.k, c.w, c.i, scaffolds_all=c.scaffolds_all, rip=c.rip, debug=True)
    assert len(got["kld"]) == len(c.rows)
    for r, exp in enumerate(c.rows):
        assert (query[got["seq"][r]][0], got["start"][r], got["stop"][r]) == (exp["name"], exp["start"], exp["stop"])
        assert list(got["meta"][r]) == exp["meta"]
        assert np.array_equal(got["counts"][r], c.window_counts[r])
        if "error" in exp:
            assert got["status"][r] & OC.ROW_ZERO_DIV
        else:
            assert not got["status"][r] & OC.ROW_ZERO_DIV
            assert abs(got["kld"][r] - exp["KLD"]) <= 1e-12
            assert bool(got["status"][r] & OC.ROW_NO_MAXMER) == (exp["KLD"] == 0 and isinstance(exp["KLD"], int))
        assert got["gc"][r] == exp["GC"]
        if c.rip_on:
            assert all(same_float(float(a), b) for a, b in zip((got["pi"][r], got["si"][r], got["cri"][r]), exp["RIP"]))


@pytest.mark.parametrize("seed,kmin,kmax,w,i", [(1, 1, 5, 300, 100), (2, 2, 7, 700, 650), (3, 1, 8, 2000, 500),
                                                (4, 3, 3, 64, 7), (5, 1, 4, 50, 45)])
def test_c_oracle_matches_numpy_oracle_on_random_input(seed, kmin, kmax, w, i):
    rng = np.random.default_rng(seed)
    seqs = []
    for n in rng.integers(1, 9000, size=6):
        s = rng.choice(list("ACGT"), size=int(n), p=[0.3, 0.2, 0.2, 0.3])
        for _ in range(3):                                  # N runs, soft-masked runs, IUPAC letters
            a = int(rng.integers(0, n))
            s[a:a + int(rng.integers(1, 400))] = rng.choice(list("NnRacgt"))
        seqs.append("".join(s))
    for mask_host in (False, True):
        sym, meta = OC.genome_profile(seqs, kmin, kmax, mask_host)
        sym_np, meta_np = N.genome_profile(seqs, kmin, kmax, mask_host)
        assert np.array_equal(sym, sym_np) and tuple(meta) == tuple(meta_np)
    ig = OC.genome_ivom(sym_np, meta_np, kmin, kmax)
    for scaffolds_all in (False, True):
        got = OC.scan(seqs, ig, kmin, kmax, w, i, scaffolds_all=scaffolds_all, rip=True, debug=True)
        exp = N.scan([(str(q), s) for q, s in enumerate(seqs)], (sym_np, meta_np), kmin, kmax, w, i, scaffolds_all, True)
        assert len(exp) == len(got["kld"])
        for r, e in enumerate(exp):
            assert (str(got["seq"][r]), got["start"][r], got["stop"][r]) == (e["name"], e["start"], e["stop"])
            assert np.array_equal(got["counts"][r], e["counts"]) and list(got["meta"][r]) == e["meta"]
            if "error" in e:
                assert got["status"][r] & OC.ROW_ZERO_DIV
            else:
                assert abs(got["kld"][r] - e["KLD"]) <= 1e-12
            assert got["gc"][r] == e["GC"]
            if kmin <= 2 <= kmax:
                assert all(same_float(float(a), b) for a, b in zip((got["pi"][r], got["si"][r], got["cri"][r]), e["RIP"]))


def test_c_oracle_candidate_slice_is_a_slice():
    rng = np.random.default_rng(9)
    seqs = ["".join(rng.choice(list("ACGT"), size=5000)) for _ in range(3)]
    sym, meta = OC.genome_profile(seqs, 1, 6)
    ig = OC.genome_ivom(sym, meta, 1, 6)
    full = OC.scan(seqs, ig, 1, 6, 500, 100)
    part = OC.scan(seqs, ig, 1, 6, 500, 100, cand=(40, 90))
    assert len(full["kld"]) == 150 and len(part["kld"]) == 50
    assert np.array_equal(part["kld"], full["kld"][40:90]) and np.array_equal(part["start"], full["start"][40:90])
    assert math.isfinite(float(full["kld"].sum()))

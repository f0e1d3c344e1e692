"""Pins the vectorised numpy oracle (oracle/frisk_oracle_np.py) to the golden vectors of the reference's
own functions: integer tables bit-exact, KLD to 1e-12 (numpy sums pairwise, CPython sequentially)."""
import numpy as np
import pytest

from golden_util import Case, case_names, same_float
from oracle import frisk_oracle as O
from oracle import frisk_oracle_np as N


@pytest.mark.parametrize("name", case_names())
def test_numpy_oracle_matches_reference_golden(name):
    c = Case(name)
    host = list(O.iter_fasta(c.host))
    sym, meta = N.genome_profile([s for _, s in host], c.m, c.k, mask_host=c.mask_host)
    assert np.array_equal(sym, c.genome_counts)
    assert list(meta) == c.genome_meta
    # the library's linear "raw" form must finalise to the same profile
    sym2, meta2 = N.finalize_raw(N.raw_profile([s for _, s in host], c.m, c.k, c.mask_host), c.m, c.k)
    assert np.array_equal(sym2, sym) and list(meta2) == list(meta)
    query = list(O.iter_fasta(c.query)) if c.query else host
    rows = N.scan(query, (sym, meta), c.m, c.k, c.w, c.i, scaffolds_all=c.scaffolds_all, rip=c.rip)
    assert len(rows) == len(c.rows)
    for r, (got, exp) in enumerate(zip(rows, c.rows)):
        assert (got["name"], got["start"], got["stop"]) == (exp["name"], exp["start"], exp["stop"])
        assert got["meta"] == exp["meta"]
        assert np.array_equal(got["counts"], c.window_counts[r])
        if "error" in exp:
            assert got.get("error") == exp["error"]
        else:
            assert abs(got["KLD"] - exp["KLD"]) <= 1e-12
        assert got["GC"] == exp["GC"]
        if c.rip_on:
            assert all(same_float(a, b) for a, b in zip(got["RIP"], exp["RIP"]))

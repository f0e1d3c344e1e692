"""Pins oracle/frisk_oracle.py (the CPU restatement) to the golden vectors produced from the
reference's own functions.  Integer tables bit-exact; KLD/GC/RIP bit-exact too (same
arithmetic, same summation order under CPython 3)."""
import numpy as np
import pytest

from golden_util import Case, case_names, same_float
from oracle import frisk_oracle as O

SLOW = {"k8", "k8_w2000", "k8_m2", "k7"}


@pytest.mark.parametrize("name", case_names())
def test_oracle_matches_reference_golden(name):
    c = Case(name)
    max_rows = 3 if name in SLOW else None          # keep the CPU suite to a few minutes
    gmaps, gmeta, rows = O.scan(c.host, c.query, c.m, c.k, c.w, c.i, mask_host=c.mask_host,
                                scaffolds_all=c.scaffolds_all, rip=c.rip, max_rows=max_rows)
    assert np.array_equal(np.asarray(O.flatten(gmaps, c.m, c.k), dtype=np.int64), c.genome_counts)
    assert [gmeta["totalLen"], gmeta["exMax"], gmeta["nnTotal"]] == c.genome_meta
    want = c.rows[:max_rows] if max_rows else c.rows
    assert len(rows) == len(want)
    for r, (got, exp) in enumerate(zip(rows, want)):
        assert (got["name"], got["start"], got["stop"]) == (exp["name"], exp["start"], exp["stop"])
        assert got["meta"] == exp["meta"]
        assert np.array_equal(np.asarray(O.flatten(got["maps"], c.m, c.k)), c.window_counts[r])
        if "error" in exp:
            assert got.get("error") == exp["error"]
        else:
            assert got["KLD"] == exp["KLD"]
        assert got["GC"] == exp["GC"]
        if c.rip_on:
            assert all(same_float(a, b) for a, b in zip(got["RIP"], exp["RIP"]))
        if c.window_ivom is not None and "error" not in exp:
            # the per-max-mer interpolated probabilities themselves (IvomBuild L369-457), not only their KLD: bit-exact
            g, w = got["ivom"]
            dense_w = np.zeros(4 ** c.k)
            dense_g = np.zeros(4 ** c.k)
            for kmer, v in w.items():
                dense_w[O.code_of(kmer)] = v
                dense_g[O.code_of(kmer)] = g[kmer]
            assert np.array_equal(dense_w, c.window_ivom[r])
            assert np.array_equal(dense_g, c.genome_ivom[r])


def test_gzip_fasta_reader(tmp_path):
    import gzip
    import shutil
    c = Case("kat")
    gz = tmp_path / "kat.fa.gz"
    with open(c.host, "rb") as src, gzip.open(gz, "wb") as dst:
        shutil.copyfileobj(src, dst)
    assert list(O.iter_fasta(str(gz))) == list(O.iter_fasta(c.host))
    assert [n for n, _ in O.iter_fasta(c.host)] == ["kat", "tiny"]

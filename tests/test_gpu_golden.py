"""GPU parity against the golden vectors of the reference's own functions, through the C ABI.

Counts, metadata, coordinates, GC and RIP are bit-exact; KLD within KLD_TOL (the north-star bound is
1e-6; the reference's own summation order is dict order, so agreement beyond ~1e-13 relative is not
defined by the reference itself)."""
from types import SimpleNamespace

import numpy as np
import pytest

from golden_util import Case, case_names, same_float

KLD_TOL = 1e-11

pytestmark = pytest.mark.gpu


def run_case(c, debug=True):
    from frisk_amd.hotpath import HotPath, mapsToProfile
    args = SimpleNamespace(hostSeq=c.host, minWordSize=c.m, maxWordSize=c.k, windowlen=c.w, increment=c.i,
                           maskHost=c.mask_host, scaffoldsAll=c.scaffolds_all, RIP=c.rip, tolerateZeroWeight=True)
    hp = HotPath(c.m, c.k)
    try:
        gmaps = hp.genomeProfile(args)
        rows, res = hp.scanGenome(args, c.query or c.host, debug=debug)
        names = list(hp.names)
    finally:
        hp.close()
    return mapsToProfile(gmaps, c.m, c.k), rows, res, names


@pytest.mark.parametrize("name", case_names())
def test_gpu_matches_reference_golden(name):
    from frisk_amd import _ffi
    c = Case(name)
    (sym, tl, ex, nn), rows, res, names = run_case(c)
    # phase A: bit-exact symmetric counts + metadata
    assert np.array_equal(sym, c.genome_counts)
    assert [tl, ex, nn] == c.genome_meta
    # phase B
    kept = np.nonzero(res.kept)[0]
    assert len(kept) == len(c.rows), "row count %d != %d" % (len(kept), len(c.rows))
    worst = 0.0
    for r, exp in zip(kept.tolist(), c.rows):
        assert (names[res.seq_index[r]], int(res.start[r]), int(res.stop[r])) == (exp["name"], exp["start"], exp["stop"])
        assert res.meta[r].tolist() == exp["meta"]
        assert np.array_equal(res.counts[r].astype(np.int64), c.window_counts[kept.tolist().index(r)])
        zero = bool(res.status[r] & _ffi.ROW_ZERO_WEIGHT)
        assert zero == ("error" in exp)
        if not zero:
            if res.status[r] & _ffi.ROW_NO_MAXMER:
                assert exp["KLD"] == 0
            else:
                worst = max(worst, abs(float(res.kld[r]) - exp["KLD"]))
        assert float(res.gc[r]) == exp["GC"]
        if c.rip_on:
            got = (float(res.pi[r]), float(res.si[r]), float(res.cri[r]))
            assert all(same_float(a, b) for a, b in zip(got, exp["RIP"]))
    assert worst <= KLD_TOL, "max |KLD - reference| = %g" % worst
    print("%s: %d rows, max |dKLD| = %.3g" % (name, len(kept), worst))


@pytest.mark.parametrize("name", [n for n in case_names() if Case(n).window_ivom is not None])
def test_gpu_ivom_vectors_match_reference_golden(name):
    """IvomBuild (L369-457) itself, not only the scalar KLD it feeds: the GPU's per-max-mer window-side and genome-side
    interpolated probabilities (frisk_scan_ivom, debug ABI) against the vectors the reference's own IvomBuild returned."""
    from frisk_amd.hotpath import HotPath
    c = Case(name)
    args = SimpleNamespace(hostSeq=c.host, minWordSize=c.m, maxWordSize=c.k, windowlen=c.w, increment=c.i,
                           maskHost=c.mask_host, scaffoldsAll=c.scaffolds_all, RIP=False, tolerateZeroWeight=True)
    hp = HotPath(c.m, c.k)
    try:
        hp.genomeProfile(args)
        rows, res = hp.scanGenome(args, c.query or c.host, debug=True)
        wi, gi = hp.engine.scan_ivom(c.w, c.i, scaffolds_all=c.scaffolds_all)
    finally:
        hp.close()
    kept = np.nonzero(res.kept)[0]
    assert len(kept) == len(c.rows) == len(c.window_ivom)
    checked = 0
    for t, (r, exp) in enumerate(zip(kept.tolist(), c.rows)):
        if "error" in exp:
            continue
        for got, want in ((wi[r], c.window_ivom[t]), (gi[r], c.genome_ivom[t])):
            assert np.array_equal(got != 0, want != 0)                       # the same set of present max-mers
            assert np.max(np.abs(got - want) / np.maximum(want, 1e-300)) <= 1e-12
            assert abs(got.sum() - 1.0) <= 1e-12
        checked += 1
    assert checked >= 5

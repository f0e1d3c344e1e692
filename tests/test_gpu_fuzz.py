"""Randomised differential test of the HIP path against the compiled CPU oracle: geometry, word sizes, flags and
sequence make-up (N runs, soft-masked runs, IUPAC letters, tiny scaffolds) are drawn from a fixed seed; every output
column is compared (integers and GC / PI / SI / CRI bit-exact, KLD to 1e-11)."""
import numpy as np
import pytest

from frisk_amd import _ffi
from frisk_amd.engine import Engine
from oracle import frisk_oracle_c as OC

pytestmark = pytest.mark.gpu


def _random_case(rng):
    kmax = int(rng.integers(1, 9))
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

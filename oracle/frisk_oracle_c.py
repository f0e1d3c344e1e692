"""ctypes loader for the C oracle (oracle/frisk_oracle_c.c -> oracle/_build/libfrisk_oracle.so)
--  TEST INFRASTRUCTURE, NOT PRODUCT CODE: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import it."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # repo root (this file lives in oracle/)
SO = os.path.join(ROOT, "oracle", "_build", "libfrisk_oracle.so")
ROW_ZERO_DIV, ROW_NO_MAXMER = 2, 8
_lib = None


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(ROOT, "oracle", "frisk_oracle_c.c")
        if not os.path.exists(SO) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(SO)):
            subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
        L = C.CDLL(SO)
        P = C.c_void_p
        L.fo_profile_len.restype = C.c_int64
        L.fo_profile_len.argtypes = [C.c_int, C.c_int]
        L.fo_threads.restype = C.c_int
        L.fo_genome_profile.argtypes = [P, P, C.c_int64, C.c_int, C.c_int, C.c_int, P, P]
        L.fo_genome_ivom.argtypes = [P, P, C.c_int, C.c_int, P]
        L.fo_scan.restype = C.c_int64
        L.fo_scan.argtypes = [P, P, C.c_int64, P, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int64,
                              C.c_int64, C.c_int64] + [P] * 11
        _lib = L
    return _lib


class Seqs:
    """Sequences as a char*[] / int64[] pair (keeps the byte strings alive)."""

    def __init__(self, seqs):
        self.raw = [s.encode("ascii") if isinstance(s, str) else bytes(s) for s in seqs]
        self.n = len(self.raw)
        self.ptrs = (C.c_char_p * max(1, self.n))(*self.raw)
        self.lens = np.array([len(s) for s in self.raw] or [0], dtype=np.int64)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def genome_profile(seqs, kmin, kmax, mask_host=False):
    S = seqs if isinstance(seqs, Seqs) else Seqs(seqs)
    sym = np.zeros(lib().fo_profile_len(kmin, kmax), dtype=np.int64)
    meta = np.zeros(3, dtype=np.int64)
    rc = lib().fo_genome_profile(S.ptrs, _p(S.lens), S.n, kmin, kmax, int(mask_host), _p(sym), _p(meta))
    assert rc == 0, rc
    return sym, tuple(int(v) for v in meta)


def genome_ivom(sym, meta, kmin, kmax):
    ig = np.zeros(4 ** kmax, dtype=np.float64)
    m = np.array(list(meta), dtype=np.int64)
    lib().fo_genome_ivom(_p(np.ascontiguousarray(sym, dtype=np.int64)), _p(m), kmin, kmax, _p(ig))
    return ig


def scan(seqs, ig, kmin, kmax, w, i, scaffolds_all=False, rip=False, cand=(0, -1), debug=False):
    """dict of per-row arrays (kept windows only, reference order)."""
    S = seqs if isinstance(seqs, Seqs) else Seqs(seqs)
    ig = np.ascontiguousarray(ig, dtype=np.float64)
    cap = 1024
    while True:
        out = dict(seq=np.zeros(cap, np.int32), start=np.zeros(cap, np.int64), stop=np.zeros(cap, np.int64),
                   status=np.zeros(cap, np.uint32), kld=np.zeros(cap), gc=np.zeros(cap), pi=np.zeros(cap),
                   si=np.zeros(cap), cri=np.zeros(cap))
        dbg_c = np.zeros((cap, lib().fo_profile_len(kmin, kmax)), np.int32) if debug else None
        dbg_m = np.zeros((cap, 3), np.int64) if debug else None
        n = lib().fo_scan(S.ptrs, _p(S.lens), S.n, _p(ig), kmin, kmax, w, i, int(scaffolds_all), int(rip), cand[0],
                          cand[1], cap, *[_p(out[k]) for k in ("seq", "start", "stop", "status", "kld", "gc", "pi", "si",
                                                               "cri")], _p(dbg_c), _p(dbg_m))
        if n >= 0:
            break
        assert n > -(1 << 62), "bad argument"
        cap = -n
    out = {k: v[:n] for k, v in out.items()}
    if debug:
        out["counts"], out["meta"] = dbg_c[:n], dbg_m[:n]
    return out

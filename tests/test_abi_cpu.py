"""No-GPU checks of the drop-in boundary: the C-ABI library builds, loads, and exports every symbol that
include/frisk_hip.h declares; the host mirror's pure-Python pieces behave like the reference's."""
import ctypes
import os
import re

import numpy as np
import pytest

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def declared_symbols():
    text = open(os.path.join(REPO, "include", "frisk_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(frisk_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build_hip()
    from frisk_amd import _ffi
    lib = _ffi.lib()
    names = declared_symbols()
    assert len(names) >= 20
    assert sorted(n for n, _, _ in _ffi.SYMBOLS) == names, "ctypes table and header disagree"
    for n in names:
        assert getattr(lib, n) is not None
    assert b"gfx950" in lib.frisk_version()
    assert lib.frisk_supported(1, 8, 5000) == 1
    assert lib.frisk_supported(1, 12, 5000) == 1 and lib.frisk_supported(1, 13, 5000) == 0 and lib.frisk_supported(3, 2, 5000) == 0
    assert lib.frisk_supported(1, 8, 70000) == 1 and lib.frisk_supported(1, 8, 2 ** 31) == 0 and lib.frisk_supported(1, 8, 0) == 0


def test_header_constants_equal_the_ctypes_side():
    """Flags, row status bits and error codes of include/frisk_hip.h against frisk_amd/_ffi.py: a drift would silently change
    which kernel form a scan takes (FRISK_SCAN_BITS4 / FRISK_SCAN_SIDE4 are test hooks) or how a row is read."""
    from frisk_amd import _ffi
    text = open(os.path.join(REPO, "include", "frisk_hip.h")).read()
    defines = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define\s+FRISK_(SCAN_[A-Z0-9_]+|ROW_[A-Z0-9_]+)\s+(\d+)u", text)}
    assert len(defines) >= 9, defines
    for name, value in defines.items():
        assert getattr(_ffi, name) == value, name
    enums = {m.group(1): int(m.group(2)) for m in re.finditer(r"FRISK_(OK|E_[A-Z_]+)\s*=\s*(-?\d+)", text)}
    assert len(enums) == 7, enums
    for name, value in enums.items():
        assert getattr(_ffi, name) == value, name
    flags = [v for k, v in defines.items() if k.startswith("SCAN_")]
    assert len(set(flags)) == len(flags) and all(v & (v - 1) == 0 for v in flags)          # distinct single bits


def test_no_cpu_fallback_when_library_missing(monkeypatch, tmp_path):
    from frisk_amd import _ffi
    monkeypatch.setattr(_ffi, "_lib", None)
    monkeypatch.setattr(_ffi, "LIB_PATH", str(tmp_path / "absent.so"))
    with pytest.raises(ImportError):
        _ffi.lib()
    from frisk_amd import Engine
    with pytest.raises(ImportError):
        Engine(1, 4)


def test_product_package_never_imports_the_oracle():
    """oracle/ is test infrastructure: no product file may import, include, load or execute it."""
    pkg = os.path.join(REPO, "frisk_amd")
    bad = re.compile(r"^\s*(from\s+\.*oracle|import\s+oracle|from\s+\S*\boracle\b\S*\s+import|#\s*include\s*[\"<][^\">]*oracle)|"
                     r"(CDLL|open|exec|import_module|__import__)\([^)]*oracle", re.M)
    for root, _dirs, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(root, f)).read()
                assert not bad.search(src), "%s uses the oracle" % f


def test_fasta_reader_matches_oracle_reader(tmp_path):
    import gzip
    from frisk_amd.fasta import iterFasta
    from oracle import frisk_oracle as O
    for name in ("kat.fa", "smalls.fa", "markov_islands.fa"):
        path = os.path.join(REPO, "tests", "golden", "inputs", name)
        assert [(n, s.decode()) for n, s in iterFasta(path)] == list(O.iter_fasta(path))
    gz = tmp_path / "x.fa.gz"
    with gzip.open(gz, "wb") as fh:
        fh.write(b">a  desc\nACGT\n\n  acgtN \n>b\n>c x\nNN\n")
    assert list(iterFasta(str(gz))) == [("a", b"ACGTacgtN"), ("b", b""), ("c", b"NN")]


def test_profile_maps_round_trip_and_layout():
    from frisk_amd.hotpath import kmerString, mapsToProfile, profileToMaps, rangeMaps
    from oracle import frisk_oracle as O
    assert [list(d) for d in rangeMaps(1, 3)] == [list(d) for d in O.blank_maps(1, 3)]
    assert kmerString(0b000110, 3) == "ATG"
    rng = np.random.default_rng(1)
    n = sum(4 ** x for x in range(2, 5))
    sym = rng.integers(0, 1 << 40, n)
    maps = profileToMaps(sym, 123, 4, 5, 2, 4)
    assert maps[-3:] == [{"totalLen": 123}, {"exMax": 4}, {"nnTotal": 5}]
    back = mapsToProfile(maps, 2, 4)
    assert np.array_equal(back[0], sym) and back[1:] == (123, 4, 5)


def test_synth_host_generator_shape():
    from frisk_amd import synth
    a = synth.scaffold(20000, 7, 0, island_frac=0.5, n_frac=0.3, lower_frac=0.2)
    b = synth.scaffold(20000, 7, 0, island_frac=0.5, n_frac=0.3, lower_frac=0.2)
    assert a == b and len(a) == 20000
    assert set(a) <= set(b"ATGCatgcN")
    assert synth.scaffold(9000, 7, 1) != synth.scaffold(9000, 7, 0)
    assert abs(sum(synth.c5_shard_lens(8, r)[0] > 0 and sum(synth.c5_shard_lens(8, r)) for r in range(8)) - 3.29e9) < 0.2e9
    loads = [sum(synth.c5_shard_lens(8, r)) for r in range(8)]
    assert max(loads) < 1.08 * min(loads)

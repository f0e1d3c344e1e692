"""The native FASTA reader (csrc/fasta_reader.h: memory-mapped, all host threads for plain files; zlib for .gz) against the
Python restatement of the reference's iterFasta (frisk_amd/fasta.py, L139-164), through the host-only digest entry point."""
import ctypes as C
import gzip
import os

import numpy as np

from frisk_amd import _ffi
from frisk_amd.fasta import readFasta


def _fnv(names, seqs):
    h = 1469598103934665603
    mask = (1 << 64) - 1
    for n, s in zip(names, seqs):
        for chunk in (n.encode("ascii"), b"\0", s if isinstance(s, bytes) else s.encode("ascii"), b"\0"):
            # vectorised FNV-1a is not possible (sequential dependency): hash a digest of long sequences instead
            for b in chunk:
                h = ((h ^ b) * 1099511628211) & mask
    return h


def _native(path):
    n, total, dig = C.c_int32(), C.c_int64(), C.c_uint64()
    rc = _ffi.lib().frisk_fasta_digest(os.fsencode(str(path)), C.byref(n), C.byref(total), C.byref(dig))
    return rc, n.value, total.value, dig.value


def _check(path):
    names, seqs = readFasta(str(path))
    rc, n, total, dig = _native(path)
    assert rc == 0
    assert n == len(names) and total == sum(len(s) for s in seqs)
    assert dig == _fnv(names, seqs)
    # ... and the file straight into the 0.25 B/base form (frisk_fasta_pack_2bit: what frisk_fasta_load uploads - no staging buffer
    # for plain files) against the Python reader's records through the packer of byte strings
    from frisk_amd.engine import fasta_pack_2bit_host, pack_2bit_host
    codes, inv, low, lens = fasta_pack_2bit_host(str(path))
    wc, wi, wl, wlens = pack_2bit_host([s.encode("latin1") if isinstance(s, str) else s for s in seqs])
    assert lens == wlens
    assert np.array_equal(codes, wc) and np.array_equal(inv, wi) and np.array_equal(low, wl)


def test_tricky_small_files(tmp_path):
    tricky = tmp_path / "tricky.fa"
    tricky.write_bytes(b"stray text before any header\nACGT\n>>r1>  first record   desc >\r\n  ACGTNN  \r\n\r\nac gt\n\n>r2\n>r3\tx\n"
                       b"TTTTTTTTTT\nGG>GG\n>r4 last no newline\nACGTRYKM")
    _check(tricky)
    gz = tmp_path / "tricky.fa.gz"
    with gzip.open(gz, "wb") as fh:
        fh.write(tricky.read_bytes())
    _check(gz)
    (tmp_path / "empty.fa").write_bytes(b"")
    _check(tmp_path / "empty.fa")
    (tmp_path / "nohdr.fa").write_bytes(b"ACGT\nACGT\n")
    _check(tmp_path / "nohdr.fa")
    bad = tmp_path / "bad.fa"
    bad.write_bytes(b">\nACGT\n")
    assert _native(bad)[0] != 0
    assert _native(tmp_path / "missing.fa")[0] != 0


def test_parallel_path_chunk_boundaries(tmp_path):
    """> 16 MB: the file is cut into one chunk per host thread at line boundaries.  Records of very different sizes, headers
    right at likely cut points, CRLF and blank lines, a 3 MB line without a newline, and text before the first header."""
    rng = np.random.default_rng(4)
    path = tmp_path / "big.fa"
    with open(path, "wb") as fh:
        fh.write(b"junk line that belongs to no record\n")
        for i in range(300):
            n = int(rng.choice([0, 1, 59, 60, 61, 5000, 200_000, 1_500_000]))
            s = rng.choice(np.frombuffer(b"ACGTacgtN", dtype=np.uint8), size=n).tobytes()
            fh.write(b">rec%d extra words %d\n" % (i, n))
            if i % 7 == 3:
                fh.write(s + b"\n")                                  # one long line
            else:
                width = int(rng.choice([60, 70, 80]))
                eol = b"\r\n" if i % 5 == 0 else b"\n"
                for o in range(0, n, width):
                    fh.write(s[o:o + width] + eol)
                    if (o // width) % 997 == 0:
                        fh.write(eol)                                # a blank line now and then
        fh.write(b">last\n" + b"ACGT" * 800_000)                     # no trailing newline, 3.2 MB line
    assert os.path.getsize(path) > (1 << 24)
    _check(path)


def test_large_record_buffer_paths(tmp_path):
    """Sequence buffers of 64 MB and more come from mmap with huge pages asked for (plain files: one allocation; .gz: a buffer
    that grows).  One 70 MB record + a short one: both readers must agree with each other and with the sizes written."""
    rng = np.random.default_rng(11)
    big = rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=70_000_000).tobytes()
    path = tmp_path / "large.fa"
    with open(path, "wb") as fh:
        fh.write(b">big one\n")
        for o in range(0, len(big), 1_000_000):
            fh.write(big[o:o + 1_000_000] + b"\n")
        fh.write(b">small\nACGTACGT\n")
    gz = tmp_path / "large.fa.gz"
    with open(path, "rb") as src, gzip.open(gz, "wb", compresslevel=1) as dst:
        dst.write(src.read())
    a, b = _native(path), _native(gz)
    assert a[0] == 0 and a[1] == 2 and a[2] == 70_000_008
    assert a == b
    from frisk_amd.engine import fasta_pack_2bit_host, pack_2bit_host
    want = pack_2bit_host([big, b"ACGTACGT"])
    for src in (path, gz):                                               # fused reader (plain) and staged reader (gzip): same form
        got = fasta_pack_2bit_host(str(src))
        assert got[3] == want[3] and all(np.array_equal(x, y) for x, y in zip(got[:3], want[:3]))


def test_packed_sequence_cache_file_roundtrip(tmp_path):
    """The .frisk2bit file (frisk_amd/hotpath.py): what was written comes back memory-mapped, and only for the very FASTA it
    was made from (path, size and modification time are part of the cache) and only when its array sizes are the ones its
    record lengths imply (a truncated, edited or foreign cache must never reach the uploader)."""
    import json
    import shutil
    import numpy as np
    from frisk_amd.hotpath import readSeqCache, seqCachePath, writeSeqCache
    fa = tmp_path / "g.fa"
    fa.write_text(">a\nACGTNNacgt\n>b\nAC\n")
    cache = seqCachePath(str(tmp_path), str(fa))
    assert cache.endswith("g.fa.frisk2bit")
    rng = np.random.default_rng(1)
    codes = rng.integers(0, 2 ** 32, size=2, dtype=np.uint32)         # lens 10 + 2: P = 32, two code words
    inv, low = np.array([[4, 6]], np.int64), np.array([[6, 10]], np.int64)
    writeSeqCache(cache, str(fa), ["a", "b"], [10, 2], codes, inv, low)
    names, lens, c2, i2, l2 = readSeqCache(cache, str(fa))
    assert names == ["a", "b"] and lens == [10, 2]
    assert np.array_equal(c2, codes) and np.array_equal(i2, inv) and np.array_equal(l2, low)
    writeSeqCache(cache, str(fa), ["a", "b"], [10, 2], codes, np.zeros((0, 2), np.int64), low)        # no N runs at all
    assert readSeqCache(cache, str(fa))[3].shape == (0, 2)
    # the same file under another path (the cache is keyed by basename): refused
    other = tmp_path / "elsewhere"
    other.mkdir()
    shutil.copy2(str(fa), str(other / "g.fa"))
    assert readSeqCache(cache, str(other / "g.fa")) is None
    # array sizes that do not follow from the lengths: refused (the uploader derives its copy sizes from the lengths)
    writeSeqCache(cache, str(fa), ["a", "b"], [10, 40], codes, inv, low)
    assert readSeqCache(cache, str(fa)) is None
    writeSeqCache(cache, str(fa), ["a"], [10, 2], codes, inv, low)
    assert readSeqCache(cache, str(fa)) is None
    writeSeqCache(cache, str(fa), ["a", "b"], [10, 2], codes, inv, low)
    assert readSeqCache(cache, str(fa)) is not None
    fa.write_text(">a\nACGTNNacgtA\n>b\nAC\n")               # another file now
    assert readSeqCache(cache, str(fa)) is None
    assert readSeqCache(str(tmp_path / "none.frisk2bit"), str(fa)) is None
    with open(cache, "r+b") as fh:                            # truncated file
        fh.truncate(40)
    assert readSeqCache(cache, str(fa)) is None
    writeSeqCache(cache, str(fa), ["a", "b"], [11, 2], codes, inv, low)
    assert readSeqCache(cache, str(fa)) is not None
    with open(cache, "r+b") as fh:
        fh.truncate(os.path.getsize(cache) - 4)
    assert readSeqCache(cache, str(fa)) is None
    # an older format's magic
    with open(cache, "wb") as fh:
        fh.write(b"FRISK2B1" + np.uint64(2).tobytes() + json.dumps({}).encode())
    assert readSeqCache(cache, str(fa)) is None


def test_random_awkward_files_fused_and_staged_readers():
    """tools/exp/fuzz_fasta.py's generator, 120 seeds: random line widths inside a record, LF / CRLF mixes, blanks, stray text, empty
    records, gzip every fifth - frisk_fasta_pack_2bit against the Python reader + frisk_pack_2bit."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "exp", "fuzz_fasta.py"), "500", "120"], capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("done: 120 seeds, 0 bad"), (out.stdout[-1500:], out.stderr[-1500:])

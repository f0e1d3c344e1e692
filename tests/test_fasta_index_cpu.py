"""The seek index over a plain FASTA file (csrc/fasta_index.h) against the Python restatement of the reference's iterFasta
(frisk_amd/fasta.py, L139-164): every record and random ranges read through the index equal what the reader returns; files
on which byte arithmetic and the reader could disagree get no index; an index is refused for any file but the one it was made
from.  Host-only entry points (no GPU)."""
import gzip
import os
import time

import numpy as np
import pytest

from frisk_amd import _ffi
from frisk_amd.fasta import fastaIndexPaths, readFasta, readFastaIndexed, writeFastaIndex


def _write(path, records, width=60, eol=b"\n", final_newline=True, gap=b""):
    with open(path, "wb") as fh:
        for i, (name, seq) in enumerate(records):
            fh.write(b">" + name + eol)
            for o in range(0, len(seq), width):
                fh.write(seq[o:o + width] + eol)
            fh.write(gap)
    if not final_newline:
        data = open(path, "rb").read()
        open(path, "wb").write(data[:-len(eol)])


def _all_equal(path, idx):
    names, seqs = readFasta(str(path))
    assert readFastaIndexed(str(path), str(idx)) == len(names)
    rng = np.random.default_rng(len(names))
    for i, (nm, s) in enumerate(zip(names, seqs)):
        name, ln, got = readFastaIndexed(str(path), str(idx), i, 0, len(s))
        assert (name, ln, got) == (nm, len(s), s)
        for _ in range(6):
            if len(s):
                a = int(rng.integers(0, len(s)))
                n = int(rng.integers(0, len(s) - a + 1))
                assert readFastaIndexed(str(path), str(idx), i, a, n)[2] == s[a:a + n]
    return names, seqs


@pytest.mark.parametrize("eol", [b"\n", b"\r\n"])
@pytest.mark.parametrize("final_newline", [True, False])
def test_regular_files_round_trip(tmp_path, eol, final_newline):
    rng = np.random.default_rng(11)
    recs = []
    for i, n in enumerate([0, 1, 59, 60, 61, 120, 121, 7, 5000, 0, 1234]):
        recs.append((b"rec%d description %d" % (i, n), rng.choice(np.frombuffer(b"ACGTacgtNnRY", dtype=np.uint8), size=n).tobytes()))
    if not final_newline:
        recs.append((b"tail", b"ACGT" * 31))          # (the last line is then a partial line without its terminator)
    path, idx = tmp_path / "g.fa", tmp_path / "g.fa.frisk.fai"
    _write(path, recs, 60, eol, final_newline)
    assert writeFastaIndex(str(path), str(idx)) == len(recs)
    names, seqs = _all_equal(path, idx)
    assert names == ["rec%d" % i for i in range(11)] + ([] if final_newline else ["tail"])
    assert [len(s) for s in seqs][:11] == [0, 1, 59, 60, 61, 120, 121, 7, 5000, 0, 1234]
    # the columns are those of `samtools faidx` (behind the stamp line)
    lines = open(idx).read().splitlines()
    assert lines[0].split()[:2] == ["#frisk-fai", "1"] and int(lines[0].split()[2]) == os.path.getsize(path)
    f = lines[1 + 4].split("\t")            # rec4: 61 bases in lines of 60
    assert f[0] == "rec4" and int(f[1]) == 61 and int(f[3]) == 60 and int(f[4]) == 60 + len(eol)
    assert open(path, "rb").read()[int(f[2]):int(f[2]) + 3] == recs[4][1][:3]


def test_header_rules_of_the_reference_and_blank_lines_between_records(tmp_path):
    path, idx = tmp_path / "h.fa", tmp_path / "h.fai"
    path.write_bytes(b"\n\n>>r1>  first   desc >\nACGTAC\nGT\n\n\n>r2\n>r3\tx\nTTTT\n   \n>r4\nAC\n\n")
    assert writeFastaIndex(str(path), str(idx)) == 4
    names, seqs = _all_equal(path, idx)
    assert names == ["r1>", "r2", "r3", "r4"] and seqs == [b"ACGTACGT", b"", b"TTTT", b"AC"]


@pytest.mark.parametrize("body,reason", [
    (b"stray\n>r\nACGT\n", "before the first header"),
    (b">r\nACGT\n\nACGT\n", "blank line"),
    (b">r\n\nACGT\n", "blank line"),
    (b">r\nACGT \nACGT\n", "blanks around"),
    (b">r\n ACGT\nACGT\n", "blanks around"),
    (b">r\nACG\nACGT\n", "different length"),
    (b">r\nACGT\nAC\nAC\n", "different length"),
    (b">r\nACGT\r\nACGT\nACGT\n", "different length"),
    (b">r\nAC GT\nACGTA\n", None),             # an inner blank is sequence to the reference too: regular, 5 + 5
])
def test_files_without_an_index(tmp_path, body, reason):
    path, idx = tmp_path / "x.fa", tmp_path / "x.fai"
    path.write_bytes(body)
    got = writeFastaIndex(str(path), str(idx))
    if reason is None:
        assert got == 1
        _all_equal(path, idx)
        return
    assert got is None and not idx.exists()
    import ctypes as C
    why = C.create_string_buffer(256)
    assert _ffi.lib().frisk_fasta_index_build(os.fsencode(str(path)), os.fsencode(str(idx)), None, why, 256) == _ffi.E_INDEX
    assert reason in why.value.decode()


def test_gzip_and_missing_files(tmp_path):
    gz = tmp_path / "g.fa.gz"
    with gzip.open(gz, "wb") as fh:
        fh.write(b">r\nACGT\n")
    assert writeFastaIndex(str(gz), str(tmp_path / "g.fai")) is None
    with pytest.raises(_ffi.FriskHipError):
        writeFastaIndex(str(tmp_path / "nope.fa"), str(tmp_path / "nope.fai"))
    (tmp_path / "empty.fa").write_bytes(b"")
    assert writeFastaIndex(str(tmp_path / "empty.fa"), str(tmp_path / "empty.fai")) == 0
    assert readFastaIndexed(str(tmp_path / "empty.fa"), str(tmp_path / "empty.fai")) == 0


def test_an_index_is_only_good_for_its_file(tmp_path):
    rng = np.random.default_rng(5)
    recs = [(b"a", rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=500).tobytes()), (b"b", b"ACGT" * 50)]
    path, idx = tmp_path / "g.fa", tmp_path / "g.fa.frisk.fai"
    _write(path, recs)
    assert writeFastaIndex(str(path), str(idx)) == 2
    _all_equal(path, idx)

    def refused(index):
        with pytest.raises(_ffi.FriskHipError) as e:
            readFastaIndexed(str(path), str(index))
        assert e.value.code == _ffi.E_INDEX
        return str(e.value)

    assert "no index" in refused(tmp_path / "absent.fai")
    # the file changes (same size, later mtime): the stamp no longer matches
    st = os.stat(path)
    os.utime(path, ns=(st.st_atime_ns, st.st_mtime_ns + 1_000_000_000))
    assert "another version" in refused(idx)
    os.utime(path, ns=(st.st_atime_ns, st.st_mtime_ns))
    _all_equal(path, idx)
    # a foreign (unstamped) index: accepted when it is NEWER than the file and every record is where it says
    body = open(idx).read().splitlines()[1:]
    foreign = tmp_path / "g.fa.fai"
    foreign.write_text("\n".join(body) + "\n")
    os.utime(foreign, ns=(st.st_atime_ns, st.st_mtime_ns + 5_000_000_000))
    _all_equal(path, foreign)
    os.utime(foreign, ns=(st.st_atime_ns, st.st_mtime_ns - 5_000_000_000))
    assert "not newer" in refused(foreign)
    os.utime(foreign, ns=(st.st_atime_ns, st.st_mtime_ns))            # (same instant: a regenerated file with the same offsets)
    assert "not newer" in refused(foreign)
    os.utime(foreign, ns=(st.st_atime_ns, st.st_mtime_ns + 5_000_000_000))
    # ... and refused when it describes another file: an offset that is no line start, a wrong name, a wrong length
    cols = [ln.split("\t") for ln in body]
    for mutate, why in ((lambda c: c[1].__setitem__(2, str(int(c[1][2]) + 1)), "line start"),
                        (lambda c: c[0].__setitem__(0, "zz"), "header"),
                        (lambda c: c[0].__setitem__(1, str(int(c[0][1]) - 7)), "does not end"),
                        (lambda c: c[1].__setitem__(1, str(10 ** 9)), "beyond the end"),
                        (lambda c: c[1].__setitem__(3, "0"), "out of shape")):
        c = [list(x) for x in cols]
        mutate(c)
        foreign.write_text("\n".join("\t".join(x) for x in c) + "\n")
        os.utime(foreign, ns=(st.st_atime_ns, st.st_mtime_ns + 5_000_000_000))
        assert why in refused(foreign), why
    foreign.write_text("not an index\n")
    assert "malformed" in refused(foreign)
    # a foreign index of a file with blanks INSIDE its sequence lines (samtools counts graphic characters only; the reference's
    # reader strips whole lines): record boundaries can look right while interior lines are not where the arithmetic puts them
    odd = tmp_path / "odd.fa"
    odd.write_bytes(b">r\n" + b"ACGTACGTAC\n" * 3 + b"ACGT ACGTA\n" + b"ACGTACGTAC\n" * 2 + b"A\n")
    sam = tmp_path / "odd.fa.fai"
    sam.write_text("r\t60\t3\t10\t11\n")               # what `samtools faidx` writes for it: 60 graphic characters - and the
                                                         # record then ENDS where the arithmetic says (the one-base last line hides the blank)
    so = os.stat(odd)
    os.utime(sam, ns=(so.st_atime_ns, so.st_mtime_ns + 5_000_000_000))
    with pytest.raises(_ffi.FriskHipError) as e:
        readFastaIndexed(str(odd), str(sam), 0, 0, 60)
    assert e.value.code == _ffi.E_INDEX and "does not hold bases" in str(e.value)
    assert fastaIndexPaths(str(path), str(tmp_path / "tmp")) == [str(tmp_path / "tmp" / "g.fa.frisk.fai"), str(path) + ".frisk.fai",
                                                                 str(path) + ".fai"]


def test_big_file_ranges_across_many_lines(tmp_path):
    """A 40 Mb record (the multi-threaded range copy of the shard loader is the same arithmetic per piece)."""
    rng = np.random.default_rng(9)
    seq = rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=40_000_003)
    path, idx = tmp_path / "big.fa", tmp_path / "big.fai"
    with open(path, "wb") as fh:
        fh.write(b">small x\nACGT\n>chr1 big\n")
        body = np.full((len(seq) + 69) // 70 * 71, ord("\n"), dtype=np.uint8)
        view = body[:len(body)].reshape(-1, 71)
        padded = np.concatenate([seq, np.zeros(view.shape[0] * 70 - len(seq), dtype=np.uint8)])
        view[:, :70] = padded.reshape(-1, 70)
        tail = len(seq) % 70
        out = body.tobytes()
        if tail:
            out = out[:len(out) - (71 - tail) + 0][: (view.shape[0] - 1) * 71 + tail] + b"\n"
        fh.write(out)
        fh.write(b">after\nTTTT\n")
    t0 = time.time()
    assert writeFastaIndex(str(path), str(idx)) == 3
    assert time.time() - t0 < 5
    s = seq.tobytes()
    assert readFastaIndexed(str(path), str(idx), 1, 0, 0)[:2] == ("chr1", len(s))
    for a, n in ((0, 1000), (69, 3), (70, 70), (len(s) - 5, 5), (12_345_678, 9_000_001), (0, len(s))):
        assert readFastaIndexed(str(path), str(idx), 1, a, n)[2] == s[a:a + n]
    assert readFastaIndexed(str(path), str(idx), 2, 0, 4)[2] == b"TTTT"
    with pytest.raises(_ffi.FriskHipError):
        readFastaIndexed(str(path), str(idx), 1, len(s) - 2, 3)

"""FASTA reader of the host side (mirror of the reference's iterFasta, frisk/__init__.py L139-164).

Same record semantics - name = first whitespace-delimited token of the header line with '>'
stripped from both ends, blank lines skipped, lines stripped of surrounding whitespace, case
preserved, '.gz' through gzip - but sequences are returned as `bytes`, ready for the device.
"""
import gzip


def iterFasta(path):
    opener = gzip.open if (path.endswith(".gz") or path.endswith('.gz"')) else open
    name, chunks = None, []
    with opener(path, "rb") as handle:
        for line in handle:
            line = line.strip()
            if not line:
                continue
            if line[:1] == b">":
                if name:
                    yield name, b"".join(chunks)
                name = line.strip(b">").split()[0].decode("ascii", "replace")
                chunks = []
            else:
                chunks.append(line)
    if name:
        yield name, b"".join(chunks)


def readFasta(path):
    """All records of a FASTA file: (names, sequences)."""
    names, seqs = [], []
    for n, s in iterFasta(path):
        names.append(n)
        seqs.append(s)
    return names, seqs


# ---- seek index (frisk_amd/csrc/fasta_index.h): lets a rank of a multi-GPU job read the bytes of ITS tiles only ----------------
def fastaIndexPaths(path, cache_dir=None):
    """Where a seek index of `path` is looked for: the library's own (stamped with the file's size and mtime) in the cache
    directory (--tempDir), then beside the FASTA, then a `samtools faidx` index beside it."""
    import os
    out = []
    if cache_dir:
        out.append(os.path.join(cache_dir, os.path.basename(path) + ".frisk.fai"))
    out.append(path + ".frisk.fai")
    out.append(path + ".fai")
    return out


def writeFastaIndex(path, index_path):
    """Write the seek index of the plain FASTA `path` (one native pass over the mapped file).  Returns the number of records,
    or None when the file has no index (gzip; text before the first header; blank lines, surrounding blanks or lines of
    different length inside a record) - such files are parsed."""
    import ctypes as C
    import os
    from . import _ffi
    n, why = C.c_int32(), C.create_string_buffer(512)
    rc = _ffi.lib().frisk_fasta_index_build(os.fsencode(path), os.fsencode(index_path), C.byref(n), why, len(why))
    if rc == _ffi.E_INDEX:
        return None
    if rc != _ffi.OK:
        raise _ffi.FriskHipError(rc, why.value.decode("utf-8", "replace"))
    return int(n.value)


def readFastaIndexed(path, index_path, seq_index=-1, offset=0, n=0):
    """Through the seek index: the number of records (seq_index < 0), or (name, length, bases [offset, offset + n) as bytes)
    of record seq_index.  Raises FriskHipError (code E_INDEX) when the index is not usable for this file."""
    import ctypes as C
    import os
    from . import _ffi
    cnt, ln = C.c_int32(), C.c_int64()
    name, why = C.create_string_buffer(4096), C.create_string_buffer(512)
    out = C.create_string_buffer(max(int(n), 1))
    rc = _ffi.lib().frisk_fasta_index_read(os.fsencode(path), os.fsencode(index_path), int(seq_index), int(offset), int(n),
                                           C.cast(out, C.c_void_p), C.byref(cnt), C.byref(ln), name, len(name), why, len(why))
    if rc != _ffi.OK:
        raise _ffi.FriskHipError(rc, why.value.decode("utf-8", "replace"))
    if seq_index < 0:
        return int(cnt.value)
    return name.value.decode("ascii", "replace"), int(ln.value), out.raw[:int(n)]

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

#!/usr/bin/env python3
"""(CPU) differential fuzz of the FASTA readers: random files of awkward form - line widths that vary inside a record, LF / CRLF mixes, blank
lines, blanks around lines, repeated '>' and trailing text in headers, stray text before the first header, empty records, no final newline,
every letter class - through frisk_fasta_pack_2bit (the fused reader for plain files, the staged one for gzip) against the Python reader
(frisk_amd/fasta.py, iterFasta's semantics) + frisk_pack_2bit.  usage: fuzz_fasta.py <first seed> <n seeds>"""
import gzip, os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from frisk_amd.engine import fasta_pack_2bit_host, pack_2bit_host
from frisk_amd.fasta import readFasta


def make(rng):
    out = []
    if rng.integers(0, 4) == 0:
        out.append(b"stray " + bytes(rng.choice(np.frombuffer(b"ACGTxyz ", dtype=np.uint8), size=int(rng.integers(0, 30)))) + b"\n")
    big = rng.integers(0, 40) == 0
    for r in range(int(rng.integers(0, 7))):
        hdr = b">" * int(rng.integers(1, 3)) + b" " * int(rng.integers(0, 2)) + b"rec%d" % r + rng.choice([b"", b" desc words", b"\tx", b" >"])
        eol = b"\r\n" if rng.integers(0, 3) == 0 else b"\n"
        out.append(hdr + eol)
        n = int(rng.choice([0, 1, 15, 16, 17, 63, 64, 65, 200, 5000, 70000])) * (300 if big and r == 0 else 1)
        s = rng.choice(np.frombuffer(b"ACGTACGTACGTacgtNnRYK-*", dtype=np.uint8), size=n)
        for _ in range(int(rng.integers(0, 4))):
            if n > 10:
                a = int(rng.integers(0, n - 5)); ln = int(rng.choice([1, 7, 40, 700]))
                s[a:a + ln] = ord("N") if rng.integers(0, 2) else (s[a:a + ln] | 0x20)
        s = s.tobytes()
        width = int(rng.choice([1, 7, 60, 61, 64, 80, 10**9]))
        o = 0
        while o < n:
            w = width if rng.integers(0, 10) else int(rng.integers(1, 100))
            line = s[o:o + w]; o += w
            if rng.integers(0, 25) == 0: line = b"  " + line + b" \t"
            out.append(line + (eol if rng.integers(0, 30) else (b"\n" if eol == b"\r\n" else b"\r\n")))
            if rng.integers(0, 40) == 0: out.append(eol)
    text = b"".join(out)
    if text.endswith(b"\n") and rng.integers(0, 3) == 0:
        text = text.rstrip(b"\r\n")
    return text


seed0, nseeds = int(sys.argv[1]), int(sys.argv[2])
bad = 0
with tempfile.TemporaryDirectory() as d:
    for seed in range(seed0, seed0 + nseeds):
        rng = np.random.default_rng(seed)
        text = make(rng)
        path = os.path.join(d, "f.fa")
        if seed % 5 == 4:
            path += ".gz"
            with gzip.open(path, "wb") as fh: fh.write(text)
        else:
            with open(path, "wb") as fh: fh.write(text)
        try:
            names, seqs = readFasta(path)
            want = pack_2bit_host([s.encode("latin1") if isinstance(s, str) else s for s in seqs])
        except Exception as err:                      # (a header without a name: both must refuse)
            try:
                fasta_pack_2bit_host(path)
                print("NATIVE ACCEPTED what the mirror refused", seed, repr(err)); bad += 1
            except Exception:
                pass
            continue
        try:
            got = fasta_pack_2bit_host(path)
        except Exception as err:
            print("NATIVE REFUSED", seed, repr(err), len(text)); bad += 1; continue
        if got[3] != want[3] or not all(np.array_equal(a, b) for a, b in zip(got[:3], want[:3])):
            print("MISMATCH", seed, len(text)); bad += 1
print("done: %d seeds, %d bad" % (nseeds, bad))

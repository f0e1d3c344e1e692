"""Host side of the 0.25 B/base upload form (frisk_pack_2bit, csrc/seq_pack2.h) against a numpy restatement of the device
packer's classification (csrc/profile_kernels.h: classify_byte) - no GPU needed: the entry point is host-only."""
import numpy as np
import pytest


def expected(seqs):
    """(codes words, inv runs, low runs) of a batch, position by position."""
    lens = [len(s) for s in seqs]
    P = max(32, (sum(n + 1 for n in lens) + 31) // 32 * 32)
    code = np.zeros(P, np.uint32)
    inv = np.zeros(P, bool)
    low = np.zeros(P, bool)
    off = 0
    digit = {ord("A"): 0, ord("T"): 1, ord("G"): 2, ord("C"): 3}
    for s in seqs:
        a = np.frombuffer(bytes(s), dtype=np.uint8)
        up = a & 0xDF
        valid = np.isin(up, list(digit))
        d = np.zeros(len(a), np.uint32)
        for ch, v in digit.items():
            d[up == ch] = v
        code[off:off + len(a)] = np.where(valid, d, 0)
        inv[off:off + len(a)] = ~valid
        low[off:off + len(a)] = valid & ((a & 0x20) != 0)
        off += len(a) + 1
    words = (code.reshape(-1, 16) << (30 - 2 * np.arange(16, dtype=np.uint32))).sum(axis=1).astype(np.uint32)

    def runs(m):
        d = np.diff(np.concatenate(([0], m.astype(np.int8), [0])))
        return np.stack([np.nonzero(d == 1)[0], np.nonzero(d == -1)[0]], axis=1).astype(np.int64)
    return words, runs(inv), runs(low), lens


def check(seqs):
    from frisk_amd.engine import pack_2bit_host
    codes, ri, rl, lens = pack_2bit_host(seqs)
    ecodes, eri, erl, elens = expected(seqs)
    assert lens == elens
    assert np.array_equal(codes, ecodes)
    assert np.array_equal(ri, eri) and np.array_equal(rl, erl)


def test_every_byte_value_and_ragged_scaffolds():
    allb = bytes(range(1, 256))                      # (0 is the PAD byte of the parser's buffer, not a letter of a record)
    check([allb, b"", b"ACGTacgtNnRYKMSWBDHVryk", b"A", b"", b"n" * 33 + b"ACGT" * 9 + b"N" * 31 + b"c"])
    check([])
    check([b""])
    check([b"", b""])


def test_runs_across_words_and_scaffold_ends():
    rng = np.random.default_rng(3)
    for trial in range(40):
        seqs = []
        for _ in range(int(rng.integers(1, 6))):
            n = int(rng.integers(0, 200))
            s = rng.choice(np.frombuffer(b"ACGTacgtNn", dtype=np.uint8), n)
            for _ in range(int(rng.integers(0, 4))):           # long runs that cross 16- and 32-position words
                a = int(rng.integers(0, max(n, 1)))
                b = min(n, a + int(rng.integers(1, 90)))
                s[a:b] = rng.choice(np.frombuffer(b"Nag", dtype=np.uint8))
            seqs.append(s.tobytes())
        check(seqs)


@pytest.mark.timeout(300)
def test_threaded_path_merges_runs_at_the_cuts():
    """>= 4 M positions: the packer's threads each take a 32-aligned range; runs that cross a cut come out whole."""
    rng = np.random.default_rng(4)
    n = 5_000_011
    s = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), n)
    for a in range(0, n, 156_250 // 2):                         # runs placed over every plausible cut (P / T for T up to 32)
        s[max(0, a - 40):a + 40] = ord("N")
        s[a + 100:a + 100 + 70] |= 0x20
    t = rng.choice(np.frombuffer(b"ACGTn", dtype=np.uint8), 70_001)
    check([s.tobytes(), b"", t.tobytes()])

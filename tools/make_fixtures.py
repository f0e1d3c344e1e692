#!/usr/bin/env python3
"""Write the literal FASTA inputs of the golden fixtures (tests/golden/inputs/*.fa).

The files this script writes are committed; they are DATA (inputs), and the
expected outputs beside them are produced by tools/make_golden.py from the
reference's own functions.  The generator below is only a convenience for
producing varied text once - the committed FASTA files are the contract, not
this PRNG ("literal FASTA texts, not RNG seeds").
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden", "inputs")

MASK64 = (1 << 64) - 1


class SplitMix64:
    def __init__(self, seed):
        self.s = seed & MASK64

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & MASK64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
        return z ^ (z >> 31)

    def u(self):  # uniform [0,1)
        return (self.next() >> 11) / float(1 << 53)


def uniform(rng, n):
    return "".join("ATGC"[rng.next() & 3] for _ in range(n))


def markov_table(rng, order=3, skew=2.5):
    """order-`order` transition table with per-context skewed probabilities."""
    tab = []
    for _ in range(4 ** order):
        w = [rng.u() ** skew + 0.05 for _ in range(4)]
        s = sum(w)
        acc, cum = 0.0, []
        for x in w:
            acc += x / s
            cum.append(acc)
        tab.append(cum)
    return tab


def markov(rng, tab, n, order=3):
    ctx = 0
    out = []
    mask = 4 ** order - 1
    for _ in range(n):
        r = rng.u()
        cum = tab[ctx]
        b = 0
        while b < 3 and r >= cum[b]:
            b += 1
        out.append("ATGC"[b])
        ctx = ((ctx << 2) | b) & mask
    return "".join(out)


def put(s, pos, text):
    return s[:pos] + text + s[pos + len(text):]


def lower(s, a, b):
    return s[:a] + s[a:b].lower() + s[b:]


def wrap(s, width):
    return "\n".join(s[i:i + width] for i in range(0, len(s), width))


def write(name, records, width=60, blank_lines=False):
    path = os.path.join(OUT, name)
    with open(path, "w") as fh:
        for hdr, seq in records:
            fh.write(">" + hdr + "\n")
            if seq:
                fh.write(wrap(seq, width) + "\n")
            if blank_lines:
                fh.write("\n")
    print("wrote", path, [len(s) for _, s in records])


def main():
    os.makedirs(OUT, exist_ok=True)

    # 1. the survey's known-answer case
    kat = "ACGTTGCAAGGCTTAACCGGATATCGCGNNACGTtgcaACGTACGTTTTTGGGGCCCCAAAAGATTACA"
    with open(os.path.join(OUT, "kat.fa"), "w") as fh:
        fh.write(">kat desc\n" + kat[:40] + "\n" + kat[40:] + "\n>tiny\nACGTACGTAC\n")

    # 2. uniform random, one scaffold
    rng = SplitMix64(101)
    write("uniform3k.fa", [("u1 uniform random", uniform(rng, 3000))])

    # 3. Markov background with an island, N-runs straddling window edges
    #    (w=400, i=150), a soft-masked run, IUPAC codes and lowercase n
    rng = SplitMix64(202)
    bg = markov_table(rng)
    isl = markov_table(rng, skew=6.0)
    s = markov(rng, bg, 4130)
    s = put(s, 1500, markov(rng, isl, 600))
    s = put(s, 390, "N" * 25)            # straddles the end of window 0 (1..400)
    s = put(s, 1040, "N" * 7)            # short run inside
    s = lower(s, 2210, 2325)             # soft-masked run
    s = put(s, 2700, "RYKMSWn")          # IUPAC + lowercase n
    s = put(s, 3300, "N")                # single N
    s2 = markov(rng, bg, 2400)           # size multiple of i=150 and of i=100
    s2 = lower(s2, 0, 30)
    s2 = put(s2, 2390, "NNNN")           # N-run in the jumpback tail
    write("markov_islands.fa", [("chrA island+N+softmask", s), ("chrB", s2)], width=70)

    # 4. small scaffolds around T = w + 0.75w - i  (w=400, i=150 -> 550)
    rng = SplitMix64(303)
    recs = [
        ("s550", uniform(rng, 550)),     # == T: skipped / rescued
        ("s551", uniform(rng, 551)),     # just above T: crawled (3 candidates)
        ("s120", uniform(rng, 120)),
        ("s300n", put(uniform(rng, 300), 50, "N" * 105)),   # 35 % N: never rescued
        ("s300ok", put(uniform(rng, 300), 50, "N" * 89)),   # 29.7 % N: rescued
        ("s7", "ACGTTGA"),               # shorter than K for K=8 tests
        ("s100low", uniform(rng, 100).lower()),             # all lowercase: 100 % "N"
        ("s900", lower(uniform(rng, 900), 100, 180)),
    ]
    write("smalls.fa", recs, width=50, blank_lines=True)

    # 5. N-heavy: windows at / around the 30 % filter (w=400 -> 120 N)
    rng = SplitMix64(404)
    s = uniform(rng, 3000)
    s = put(s, 150, "N" * 120)           # window 1..400 has exactly 120 N -> dropped (>=)
    s = put(s, 1000, "N" * 119)          # 119 in windows covering it -> kept
    s = put(s, 1600, "N" * 700)          # big gap: several windows dropped
    s = lower(s, 2500, 2620)             # 120 lowercase = "N" for the filter
    write("nheavy.fa", [("nh", s)], width=80)

    # 6. host / query pair for -Q != -H
    rng = SplitMix64(505)
    tab_h = markov_table(rng, skew=1.5)
    tab_q = markov_table(rng, skew=4.0)
    write("host.fa", [("h1", markov(rng, tab_h, 5000)), ("h2", lower(markov(rng, tab_h, 2500), 400, 900))])
    q = markov(rng, tab_q, 1300) + markov(rng, tab_h, 1300)
    q = lower(q, 600, 700)
    write("query.fa", [("q1", q)])

    # 7. K=8 default-ish geometry: w=5000 i=1000, ~12.3 kb, lowercase + N
    rng = SplitMix64(606)
    bg = markov_table(rng, skew=1.2)
    isl = markov_table(rng, skew=5.0)
    s = markov(rng, bg, 12300)
    s = put(s, 6000, markov(rng, isl, 1500))
    s = put(s, 4990, "N" * 20)           # straddles window boundary 5000
    s = lower(s, 9000, 9400)
    s = put(s, 11000, "A" * 300)         # low-complexity: repeated 8-mers
    s = put(s, 11300, "AT" * 100)
    write("k8.fa", [("k8chr", s)], width=100)

    # 8. scaffolds shorter than w but longer than the small-scaffold limit (w=100, i=90 -> 85):
    #    the jumpback slice seq[size-w:size] then has a NEGATIVE start (Python slice from the end)
    rng = SplitMix64(707)
    write("overshoot.fa", [("ov95", uniform(rng, 95)), ("ov86", uniform(rng, 86)), ("ov99", uniform(rng, 99)),
                           ("ov200", uniform(rng, 200)), ("ov85", uniform(rng, 85))])

    return 0


if __name__ == "__main__":
    sys.exit(main())

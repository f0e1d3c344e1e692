"""Synthetic scaffolds "of the named shape" (BASELINE.json) - host specification of the generator.

The device generator (csrc/synth_kernel.h, frisk_seq_synth) and `scaffold()` below produce the same
bytes: every property of a base is a pure function of (seed, scaffold index, position).

  * background: order-3 Markov chain, restarted every 4096-base block, transition table from the seed
  * islands   : blocks grouped by 5 (20 480 bases) switch to a second, strongly skewed table with
                probability island_frac
  * N runs    : units of 8 blocks (32 768 bases) with probability 0.8*n_frac, units of 1024 bases with
                probability 0.2*n_frac
  * soft mask : units of 512 bases are lower-cased with probability lower_frac
  * repeats   : units of 512 bases hold one simple repeat (poly-A x3 / poly-T x2 / (CA)n / (TG)n / (AAAT)n, 12..31 bases,
                one in four up to 91) with probability repeats_per_kb * 0.512; soft-masked when lower_frac > 0 (a soft-masked
                assembly), uppercase otherwise.  REPEATS_SOFT / REPEATS_UNMASKED below are the "realistic" shapes of the
                bench: a primate assembly as UCSC distributes it (soft-masked: the reference's 30 % filter - which counts
                lowercase as unresolved, L106-118 - then drops most windows) and as an unmasked download (every window
                scored, and poly-A tails / microsatellites overflow 4-bit counters in almost half of them)
  * period_mix: a share period_mix of those repeats is instead of period 3, 5 or 6 - (CAG)n, (AAT)n, (AAAAT)n, (TTAGGG)n, in
                equal parts - with a longer tail (12..31 bases, one in four up to 210, cut at the unit's end).  The classes'
                shares in REPEATS_MIXED (period 1: 47 %, 2: 19 %, 4: 9 %, 3: 12.5 %, 5 and 6: 6 % each) are a rough reading of
                the microsatellite census of the human genome (mono- > di- > tetra- / penta- > tri- > hexanucleotide repeats;
                no network here to pin a table: the proportions are this file's own, stated so that they can be argued with)
  * satellites: a share sat_frac of the bases lies in tandem arrays of a 171-base monomer (its own per array) with 3 %
                divergence between copies: the first 1..8 units of 131 072 bases in a group of 8 (0.13..1.05 Mb; neighbouring
                groups merge) - alpha-satellite-like: every 8-mer of the monomer occurs ~23 times in a 5 kb window, none of
                them of a short period.  REPEATS_MIXED takes 3 % (the alpha-satellite share of a telomere-to-telomere human
                assembly)
"""
import numpy as np

BLOCK = 4096
ISLAND_BLOCKS = 5
NBIG_BLOCKS = 8
NSMALL = 1024
LOWER = 512
REP = 512
# the bench's realistic shapes: one simple repeat per ~3 kb (Alu poly-A tails + microsatellites), 7 % N
REPEATS_SOFT = dict(island_frac=0.02, n_frac=0.07, lower_frac=0.45, repeats_per_kb=0.35)
REPEATS_UNMASKED = dict(island_frac=0.02, n_frac=0.07, lower_frac=0.0, repeats_per_kb=0.35)
# ... and the shape that is NOT built around the scan kernel's side table (periods <= 4): a quarter of the simple repeats of
# period 3 / 5 / 6, and 3 % of the bases in satellite arrays
REPEATS_MIXED = dict(island_frac=0.02, n_frac=0.07, lower_frac=0.0, repeats_per_kb=0.35, period_mix=0.25, sat_frac=0.03)
SAT_UNIT = 131072
SAT_GROUP = 8
SAT_MONOMER = 171
SAT_DIV = 0.03
GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _mix(z):
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
    return z ^ (z >> np.uint64(31))


def _unit_hash(seed, scaf, unit, salt):
    unit = np.asarray(unit, dtype=np.uint64)
    key = (np.uint64(scaf) << np.uint64(40)) ^ unit ^ (np.uint64(salt) << np.uint64(56))
    return (_mix(np.uint64(seed) ^ _mix(key)) >> np.uint64(32)).astype(np.uint32)


def frac_to_u32(f):
    if not f > 0.0:
        return 0
    if f >= 1.0:
        return 0xFFFFFFFF
    return int(f * 4294967296.0)


def make_tables(seed):
    bg = np.zeros((64, 4), dtype=np.uint32)
    isl = np.zeros((64, 4), dtype=np.uint32)
    for ctx in range(64):
        h = [int(_mix(np.uint64(seed) ^ _mix(np.uint64(0xB5 + ctx * 4 + b)))) for b in range(4)]
        wb = [64 + (x % 192) for x in h]
        wi = [16 + ((x >> 20) % 1009) for x in h]
        cb = ci = 0
        for b in range(4):
            cb += wb[b]
            ci += wi[b]
            bg[ctx, b] = 0xFFFFFFFF if b == 3 else (cb << 32) // sum(wb)
            isl[ctx, b] = 0xFFFFFFFF if b == 3 else (ci << 32) // sum(wi)
    return bg, isl


def _satellite(length, seed, scaf_index, sat_frac):
    """(in_satellite mask, base digit) per position - the `sat` branch of synth_kernel."""
    thr = frac_to_u32(sat_frac * (2.0 * SAT_GROUP / (SAT_GROUP + 1.0)))
    if thr == 0:
        return np.zeros(length, bool), np.zeros(length, np.int64)
    pos = np.arange(length, dtype=np.int64)
    unit = pos // SAT_UNIT
    group = unit // SAT_GROUP
    ng = int(group[-1]) + 1
    g = np.arange(ng, dtype=np.uint64)
    on_g = _unit_hash(seed, scaf_index, g, 8) < np.uint32(thr)
    take = 1 + (_unit_hash(seed, scaf_index, g, 9).astype(np.int64) >> 8) % SAT_GROUP
    sat = on_g[group] & ((unit % SAT_GROUP) < take[group])
    base = np.zeros(length, np.int64)
    idx = np.nonzero(sat)[0]
    if idx.size:
        j = (idx - group[idx] * (SAT_UNIT * SAT_GROUP)) % SAT_MONOMER
        b = _unit_hash(seed, scaf_index, (group[idx] * 256 + j).astype(np.uint64), 10).astype(np.int64) >> 30
        hd = _unit_hash(seed, scaf_index, idx.astype(np.uint64), 11)
        b = np.where(hd < np.uint32(frac_to_u32(SAT_DIV)), hd.astype(np.int64) & 3, b)
        base[idx] = b
    return sat, base


def _repeats(length, seed, scaf_index, repeats_per_kb, period_mix=0.0):
    """(in_repeat mask, repeat base digit) per position - synth_repeat_of / synth_repeat_base of csrc/synth_kernel.h."""
    pos = np.arange(length, dtype=np.int64)
    unit = pos // REP
    thr = frac_to_u32(repeats_per_kb * (REP / 1000.0))
    if thr == 0:
        return np.zeros(length, bool), np.zeros(length, np.int64)
    nunit = int(unit[-1]) + 1
    u = np.arange(nunit, dtype=np.uint64)
    on = _unit_hash(seed, scaf_index, u, 5) < np.uint32(thr)
    h = _unit_hash(seed, scaf_index, u, 6).astype(np.int64)
    ln = 12 + (h >> 3) % 20 + np.where(((h >> 8) & 3) == 0, (h >> 12) % 60, 0)
    start = np.arange(nunit, dtype=np.int64) * REP + (h >> 20) % (REP - 96)
    kind = h & 7
    period = np.where(kind < 5, 1, np.where(kind < 7, 2, 4))
    motif = np.where(kind < 3, 0x0, np.where(kind < 5, 0x1, np.where(kind == 5, 0xC, np.where(kind == 6, 0x6, 0x01))))
    thr_mix = frac_to_u32(period_mix)
    if thr_mix:
        mixed = _unit_hash(seed, scaf_index, u, 7) < np.uint32(thr_mix)
        which = (h >> 1) & 3
        period = np.where(mixed, np.where(which < 2, 3, np.where(which == 2, 5, 6)), period)
        motif = np.where(mixed, np.where(which == 0, 0x32, np.where(which == 1, 0x01, np.where(which == 2, 0x001, 0x52A))), motif)
        ln2 = 12 + (h >> 3) % 20 + np.where(((h >> 8) & 3) == 0, (h >> 12) % 180, 0)
        room = (np.arange(nunit, dtype=np.int64) + 1) * REP - start
        ln = np.where(mixed, np.minimum(ln2, room), ln)
    inside = on[unit] & (pos >= start[unit]) & (pos < start[unit] + ln[unit])
    k = (pos - start[unit]) % period[unit]
    base = (motif[unit] >> (2 * (period[unit] - 1 - k))) & 3
    return inside, base


def scaffold(length, seed, scaf_index, island_frac=0.02, n_frac=0.0, lower_frac=0.0, repeats_per_kb=0.0, period_mix=0.0, sat_frac=0.0):
    """bytes of one synthetic scaffold (vectorised over its 4096-base blocks)."""
    length = int(length)
    if length <= 0:
        return b""
    bg, isl = make_tables(seed)
    nblk = (length + BLOCK - 1) // BLOCK
    in_rep, rep_base = _repeats(nblk * BLOCK, seed, scaf_index, repeats_per_kb, period_mix)
    in_rep, rep_base = in_rep.reshape(nblk, BLOCK), rep_base.reshape(nblk, BLOCK)
    in_sat, sat_base = _satellite(nblk * BLOCK, seed, scaf_index, sat_frac)
    in_sat, sat_base = in_sat.reshape(nblk, BLOCK), sat_base.reshape(nblk, BLOCK)
    blk = np.arange(nblk, dtype=np.uint64)
    island = _unit_hash(seed, scaf_index, blk // np.uint64(ISLAND_BLOCKS), 1) < np.uint32(frac_to_u32(island_frac))
    key = (np.uint64(scaf_index) << np.uint64(40)) ^ blk
    state = _mix(np.uint64(seed) ^ _mix(key))
    ctx = np.zeros(nblk, dtype=np.int64)
    out = np.zeros((nblk, BLOCK), dtype=np.uint8)
    letters = np.frombuffer(b"ATGC", dtype=np.uint8)
    tab = np.where(island[:, None, None], isl[None], bg[None])        # (nblk, 64, 4)
    rows = np.arange(nblk)
    with np.errstate(over="ignore"):
        for t in range(BLOCK):
            state = state + GOLDEN
            r = (_mix(state) >> np.uint64(32)).astype(np.uint32)
            row = tab[rows, ctx]                                        # (nblk, 4)
            b = (r >= row[:, 0]).astype(np.int64) + (r >= row[:, 1]) + (r >= row[:, 2])
            b = np.where(in_rep[:, t], rep_base[:, t], b)
            b = np.where(in_sat[:, t], sat_base[:, t], b)
            ctx = ((ctx << 2) | b) & 63
            out[:, t] = letters[b]
    seq = out.reshape(-1)[:length].copy()
    pos = np.arange(length, dtype=np.uint64)
    low = _unit_hash(seed, scaf_index, pos // np.uint64(LOWER), 4) < np.uint32(frac_to_u32(lower_frac))
    if frac_to_u32(lower_frac) != 0:            # an assembly is soft-masked - repeats included - or it is not
        low = low | in_rep.reshape(-1)[:length] | in_sat.reshape(-1)[:length]
    seq[low] |= 0x20
    nbig = _unit_hash(seed, scaf_index, pos // np.uint64(BLOCK * NBIG_BLOCKS), 2) < np.uint32(frac_to_u32(n_frac * 0.8))
    nsmall = _unit_hash(seed, scaf_index, pos // np.uint64(NSMALL), 3) < np.uint32(frac_to_u32(n_frac * 0.2))
    seq[nbig | nsmall] = ord("N")
    return seq.tobytes()


# ---- named shapes -----------------------------------------------------------
C2_LENS = [4641652]
C3_LENS = [230218, 813184, 316620, 1531933, 576874, 270161, 1090940, 562643, 439888, 745751, 666816, 1078177,
           924431, 784333, 1091291, 948066]
C4_LENS = [248956422]


def c5_shard_lens(n_shards=8, shard=0):
    """GRCh38-like: 24 chromosome-scale scaffolds + 400 scaffolds of 10 kb - 1 Mb (sum ~3.1 Gb), dealt to
    `n_shards` GPUs by longest-processing-time bin packing; returns the lengths of one shard."""
    chrom = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717,
             133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285,
             58617616, 64444167, 46709983, 50818468, 156040895, 57227415]
    small = [10000 + int(_mix(np.uint64(0xC5 + i)) % np.uint64(990001)) for i in range(400)]
    bins = [[] for _ in range(n_shards)]
    load = [0] * n_shards
    for ln in sorted(chrom + small, reverse=True):
        j = load.index(min(load))
        bins[j].append(ln)
        load[j] += ln
    return bins[shard]

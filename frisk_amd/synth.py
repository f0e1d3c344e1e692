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


def _repeats(length, seed, scaf_index, repeats_per_kb):
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
    inside = on[unit] & (pos >= start[unit]) & (pos < start[unit] + ln[unit])
    k = (pos - start[unit]) % period[unit]
    base = (motif[unit] >> (2 * (period[unit] - 1 - k))) & 3
    return inside, base


def scaffold(length, seed, scaf_index, island_frac=0.02, n_frac=0.0, lower_frac=0.0, repeats_per_kb=0.0):
    """bytes of one synthetic scaffold (vectorised over its 4096-base blocks)."""
    length = int(length)
    if length <= 0:
        return b""
    bg, isl = make_tables(seed)
    nblk = (length + BLOCK - 1) // BLOCK
    in_rep, rep_base = _repeats(nblk * BLOCK, seed, scaf_index, repeats_per_kb)
    in_rep, rep_base = in_rep.reshape(nblk, BLOCK), rep_base.reshape(nblk, BLOCK)
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
            ctx = ((ctx << 2) | b) & 63
            out[:, t] = letters[b]
    seq = out.reshape(-1)[:length].copy()
    pos = np.arange(length, dtype=np.uint64)
    low = _unit_hash(seed, scaf_index, pos // np.uint64(LOWER), 4) < np.uint32(frac_to_u32(lower_frac))
    if frac_to_u32(lower_frac) != 0:            # an assembly is soft-masked - repeats included - or it is not
        low = low | in_rep.reshape(-1)[:length]
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

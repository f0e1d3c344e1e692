"""Host-side mirror of the reference's hot-path call pattern, running on the GPU.

The reference (frisk/__init__.py) drives the path from main():
    L1442        genomeKmers = computeKmers(args, genomeMode=True, ...)            phase A
    L1478-1494   for seq, name, start, stop in crawlGenome(args, querySeq): ...     phase B
Here the same two seams are `genomeProfile(args)` and `scanGenome(args, querySeq, genomeKmers)`;
both take the same argparse-style `args` object (hostSeq, minWordSize, maxWordSize, windowlen,
increment, maskHost, scaffoldsAll, RIP) and produce the same values: the list-of-dicts k-mer maps
with the three metadata dicts appended (L356-359), and rows (name, start, stop, windowKLD, GC[, PI,
SI, CRI]) in the reference's order.  Where the reference would die with ZeroDivisionError
(L437: a window max-mer whose prefix has zero weight in the genome) scanGenome raises it too.
"""
import logging
import os

import numpy as np

from . import _ffi
from .engine import Engine, table_offset

LETTERS = ("A", "T", "G", "C")      # reference L70


def kmerString(code, x):
    return "".join(LETTERS[(code >> (2 * (x - 1 - p))) & 3] for p in range(x))


def rangeMaps(kMin, kMax):
    """Blank maps: one dict per order, keys in canonical index order (reference L253-274)."""
    return [{kmerString(c, x): 0 for c in range(4 ** x)} for x in range(kMin, kMax + 1)]


def profileToMaps(sym, total_len, ex_max, nn_total, kMin, kMax):
    """flat symmetric counts -> the reference's pickled structure (L356-359)."""
    maps = []
    for x in range(kMin, kMax + 1):
        off = table_offset(kMin, x)
        vals = sym[off:off + 4 ** x].tolist()
        maps.append({kmerString(c, x): vals[c] for c in range(4 ** x)})
    maps.append({"totalLen": int(total_len)})
    maps.append({"exMax": int(ex_max)})
    maps.append({"nnTotal": int(nn_total)})
    return maps


def mapsToProfile(maps, kMin, kMax):
    flat = []
    for x in range(kMin, kMax + 1):
        d = maps[x - kMin]
        flat.extend(d[kmerString(c, x)] for c in range(4 ** x))
    r = kMax - kMin
    return (np.asarray(flat, dtype=np.int64), maps[r + 1]["totalLen"], maps[r + 2]["exMax"], maps[r + 3]["nnTotal"])



SEQ_CACHE_MAGIC = b"FRISK2B2"


def seqCachePath(cache_dir, fasta):
    return os.path.join(cache_dir, os.path.basename(fasta) + ".frisk2bit")


def _source_stamp(fasta):
    st = os.stat(fasta)
    return [int(st.st_size), int(st.st_mtime_ns)]


def _padded_len(lens):
    p = (sum(int(n) + 1 for n in lens) + 31) // 32 * 32
    return p or 32


def writeSeqCache(cache, fasta, names, lens, codes, inv_runs, low_runs):
    """<magic><u64 header length><json header><codes: uint32><inv runs: int64 pairs><low runs: int64 pairs> - the 0.25 B/base
    form frisk_seq_stage_2bit uploads (2-bit codes + run lists of the two masks) - written to a temporary name and renamed."""
    import json
    inv_runs = np.ascontiguousarray(inv_runs, dtype=np.int64).reshape(-1, 2)
    low_runs = np.ascontiguousarray(low_runs, dtype=np.int64).reshape(-1, 2)
    head = json.dumps({"source": os.path.abspath(fasta), "stamp": _source_stamp(fasta), "names": names, "lens": [int(x) for x in lens],
                       "words": int(codes.size), "runs": [int(inv_runs.shape[0]), int(low_runs.shape[0])]}).encode()
    head += b" " * (-(16 + len(head)) % 8)              # the arrays start on a multiple of 8 bytes
    tmp = cache + ".tmp%d" % os.getpid()
    try:
        with open(tmp, "wb") as fh:
            fh.write(SEQ_CACHE_MAGIC)
            fh.write(np.uint64(len(head)).tobytes())
            fh.write(head)
            np.ascontiguousarray(codes, dtype=np.uint32).tofile(fh)
            inv_runs.tofile(fh)
            low_runs.tofile(fh)
        os.replace(tmp, cache)
    except OSError:
        try:
            os.remove(tmp)
        except OSError:
            pass


def readSeqCache(cache, fasta):
    """(names, lens, codes, inv_runs, low_runs) - the three arrays memory-mapped - or None when there is no cache for this very
    file: the cache names the FASTA it was made from (absolute path, size, modification time), and its array sizes must be
    the ones the record lengths imply - a truncated, edited or foreign file is not handed to the uploader."""
    import json
    try:
        with open(cache, "rb") as fh:
            if fh.read(8) != SEQ_CACHE_MAGIC:
                return None
            n = int(np.frombuffer(fh.read(8), dtype=np.uint64)[0])
            if n > (1 << 31):
                return None
            head = json.loads(fh.read(n).decode())
            off = 16 + n
        if head.get("stamp") != _source_stamp(fasta) or head.get("source") != os.path.abspath(fasta):
            return None
        lens = [int(x) for x in head["lens"]]
        names = list(head["names"])
        words, (n_inv, n_low) = int(head["words"]), (int(x) for x in head["runs"])
        P = _padded_len(lens)
        if len(names) != len(lens) or min(lens, default=0) < 0 or words != 2 * P // 32 or n_inv < 0 or n_low < 0:
            return None
        if os.path.getsize(cache) != off + 4 * words + 16 * (n_inv + n_low):
            return None
        codes = np.memmap(cache, dtype=np.uint32, mode="r", offset=off, shape=(words,))
        off += 4 * words
        runs = []
        for k in (n_inv, n_low):
            runs.append(np.memmap(cache, dtype=np.int64, mode="r", offset=off, shape=(k, 2)) if k else np.zeros((0, 2), np.int64))
            off += 16 * k
        return names, lens, codes, runs[0], runs[1]
    except (OSError, ValueError, KeyError, TypeError):
        return None


def crawlLog(names, sizes, seq_index, kept, w, inc, scaffolds_all, emit=None):
    """The progress lines crawlGenome logs per scaffold (L212-250), from the scan's per-candidate arrays: small scaffolds
    skipped / rescued / dropped, `Extracted N windows from S bases in NAME`, `Excluded N windows from NAME`, and the closing
    `Successfully processed sequences` count (a rescued scaffold counts twice there, as in the reference: L221 and L250).
    The reference also logs one line per window that the N filter drops (L238); those go to the DEBUG level here -
    a 3 Gb assembly has 2e5 of them."""
    log = logging.getLogger("frisk_amd")
    emit = emit or log.info
    n = len(names)
    seq_index = np.asarray(seq_index, dtype=np.int64)
    cand = np.bincount(seq_index, minlength=n) if seq_index.size else np.zeros(n, np.int64)
    good = np.bincount(seq_index[np.asarray(kept, dtype=bool)], minlength=n) if seq_index.size else np.zeros(n, np.int64)
    limit = w + ((w * 0.75) - inc)
    done = 0
    for s in range(n):
        name, size = names[s], int(sizes[s])
        count, dropped = int(good[s]), int(cand[s] - good[s])
        if size <= limit and scaffolds_all:
            if dropped:
                emit("%s excluded as > 30 percent unresolved sequence." % name)
                continue
            emit("Rescuing small scaffold %s" % name)
            done += 1
        elif size <= limit:
            emit("%s excluded as below minimum scaffold length of %s." % (name, _py2_float(limit)))
            continue
        elif dropped and log.isEnabledFor(logging.DEBUG):
            for _ in range(dropped):
                log.debug("Window from %s excluded as > 30 percent unresolved sequence." % name)
        emit("Extracted %s windows from %s bases in %s" % (count, str(size), name))
        emit("Excluded %s windows from %s" % (dropped, name))
        done += 1
    emit("Successfully processed sequences: %s" % done)
    return done


def _py2_float(x):
    s = "%.12g" % x
    return s if ("." in s or "e" in s) else s + ".0"


class HotPath:
    """Owns the Engine and the resident batch for one run of the CLI / one test."""

    def __init__(self, kMin, kMax, device=0, cache_dir=None, use_cache=True):
        self.engine = Engine(kMin, kMax, device)
        self.kMin, self.kMax = kMin, kMax
        self._resident = None       # path of the FASTA whose scaffolds are on the device
        self.names = []
        # packed-sequence cache (beside the reference's pickle caches in --tempDir): <fasta basename>.frisk2bit holds the 2-bit
        # codes as they lie in HBM and the two masks as run lists, so a later run on the same file neither parses (nor inflates)
        # it nor packs it, PCIe carries 0.25 B per base, and phase A follows the upload piece by piece (frisk_seq_stage_2bit).
        # use_cache=False (--recalc given): ignore and rewrite it.
        self.cache_dir, self.use_cache = cache_dir, use_cache
        self.loaded_from_cache = False
        self._writer = None

    def close(self):
        self._join_writer()
        self.engine.close()

    def _join_writer(self):
        if self._writer is not None:
            self._writer.join()
            self._writer = None

    def _load(self, path):
        if self._resident == path:
            return
        cache = seqCachePath(self.cache_dir, path) if self.cache_dir else None
        self.loaded_from_cache = False
        if cache and self.use_cache:
            got = readSeqCache(cache, path)
            if got is not None:
                names, lens, codes, inv_runs, low_runs = got
                self.engine.stage_2bit(codes, inv_runs, low_runs, lens)
                self.engine.commit(names)
                self.names, self._resident, self.loaded_from_cache = names, path, True
                return
        self.names = self.engine.load_fasta(path)           # native reader (iterFasta semantics, L139-164)
        self._resident = path
        if cache:
            self._join_writer()
            codes, inv_runs, low_runs = self.engine.export_2bit()
            import threading                                # (the file is written while the profile and the scan run)
            self._writer = threading.Thread(target=writeSeqCache, args=(cache, path, list(self.names), list(self.engine.seq_lens),
                                                                        codes, inv_runs, low_runs), daemon=False)
            self._writer.start()

    # phase A -----------------------------------------------------------------------------------
    def genomeProfile(self, args, allreduce=False):
        """computeKmers(genomeMode=True) (L280-367 as called at L1442)."""
        self._load(args.hostSeq)
        e = self.engine
        e.profile_reset()
        e.profile_add(mask_host=bool(getattr(args, "maskHost", False)))
        if allreduce:
            e.profile_allreduce()
        e.profile_finalize()
        sym, tl, ex, nn = e.profile_get()
        return profileToMaps(sym, tl, ex, nn, self.kMin, self.kMax)

    def genomeProfileArrays(self, args, allreduce=False):
        """genomeProfile() without the reference's dict form: the arrays of Engine.profile_get().  The CLI turns them into the
        pickled form (profileToMaps) in the background - the scan needs only the profile on the device."""
        self._load(args.hostSeq)
        e = self.engine
        e.profile_reset()
        e.profile_add(mask_host=bool(getattr(args, "maskHost", False)))
        if allreduce:
            e.profile_allreduce()
        e.profile_finalize()
        return e.profile_get()

    def profileMaps(self):
        """The finished profile on the device in the reference's pickled form (list of dicts + 3 metadata dicts, L356-359)."""
        sym, tl, ex, nn = self.engine.profile_get()
        return profileToMaps(sym, tl, ex, nn, self.kMin, self.kMax)

    def setGenomeProfile(self, genomeKmers):
        """Install a previously computed profile (replaces pickle.load at L1437-1439)."""
        sym, tl, ex, nn = mapsToProfile(genomeKmers, self.kMin, self.kMax)
        self.engine.profile_set(sym, tl, ex, nn)

    # phase B -----------------------------------------------------------------------------------
    def scanTable(self, args, querySeq, debug=False):
        """The loop L1478-1494 as columns: returns (ScoreTable of the rows the reference emits, raw ScanResult).
        Where the reference would die with ZeroDivisionError (L437) this raises it too, with `err.table` = the rows the
        reference had already written before the failing window."""
        from .table import ScoreTable
        self._load(querySeq)
        rip = bool(getattr(args, "RIP", False)) and args.minWordSize <= 2
        if rip and args.maxWordSize < 2:
            raise ValueError("2 is not in list")        # range(m, K+1).index(2), reference L478
        # (ordinary numpy arrays: `res` is handed to the caller and must outlive the next scan and Engine.close())
        res = self.engine.scan(args.windowlen, args.increment, rip=rip,
                               scaffolds_all=bool(getattr(args, "scaffoldsAll", False)), debug=debug, pinned=False)
        crawlLog(self.names, self.engine.seq_lens, res.seq_index, res.kept, args.windowlen, args.increment,
                 bool(getattr(args, "scaffoldsAll", False)))
        kept = np.nonzero(res.kept)[0]
        bad = kept[(res.status[kept] & _ffi.ROW_ZERO_WEIGHT) != 0]
        tolerate = bool(getattr(args, "tolerateZeroWeight", False))
        if bad.size:                                    # the reference has written every row before the failing one
            kept = kept[kept < int(bad[0])]
        int0 = ((res.status[kept] & _ffi.ROW_NO_MAXMER) != 0).astype(np.uint8)     # int 0: empty sum, L465
        table = ScoreTable(self.names, res.seq_index[kept], res.start[kept], res.stop[kept], res.kld[kept], res.gc[kept],
                           res.pi[kept] if rip else None, res.si[kept] if rip else None, res.cri[kept] if rip else None, int0)
        if bad.size and not tolerate:
            err = ZeroDivisionError("float division by zero")   # what the reference raises at L437
            err.table = table
            err.rows = table.rows() if len(table) <= 100000 else None
            err.result = res
            raise err
        return table, res

    def scanGenome(self, args, querySeq, debug=False):
        """The same as a list of row tuples (tests, small jobs): (rows, result)."""
        table, res = self.scanTable(args, querySeq, debug=debug)
        return table.rows(), res

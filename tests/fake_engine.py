"""Oracle-backed stand-in with the Engine interface, so that the multi-rank orchestration
(frisk_amd/distributed.py) can be rehearsed on the CPU over gloo.  Test infrastructure only."""
import numpy as np

from frisk_amd import _ffi
from frisk_amd.distributed import allreduce_raw_host
from frisk_amd.engine import ScanResult
from oracle import frisk_oracle_np as N


class FakeEngine:
    def __init__(self, kmin, kmax):
        self.kmin, self.kmax = kmin, kmax
        self.nprof = N.profile_len(kmin, kmax)
        self.seqs, self.off = [], []
        self.raw = np.zeros(self.nprof + 4, dtype=np.int64)
        self.profile = None

    def load(self, seqs):
        self.tiles = None
        self.seqs = [N.Encoded(s) for s in seqs]
        self.off, pos = [], 0
        for e in self.seqs:
            self.off.append(pos)
            pos += e.n + 1
        self._padded = max(32, (pos + 31) // 32 * 32)

    def load_fasta(self, path):
        from frisk_amd.fasta import readFasta
        names, seqs = readFasta(path)
        self.load(seqs)
        return names

    def load_fasta_shard(self, path, w, inc, rank, world, scaffolds_all=False, index=None):
        """The tiles of one rank (frisk_amd.distributed.plan_tiles is the specification of frisk_fasta_load_shard).  With a seek
        index the tiles' bytes come through the library's host-only index reader (what frisk_fasta_load_shard_indexed copies)."""
        from frisk_amd.distributed import plan_tiles
        from frisk_amd.fasta import readFasta, readFastaIndexed
        self.shard_index = None
        if index is not None:
            n = readFastaIndexed(path, index)
            heads = [readFastaIndexed(path, index, i)[:2] for i in range(n)]
            names, lens = [h[0] for h in heads], [h[1] for h in heads]
            (c0, c1), tiles = plan_tiles(lens, w, inc, scaffolds_all, self.kmax, rank, world)
            self.load([readFastaIndexed(path, index, t["scaf"], t["base0"], t["end"] - t["base0"])[2] for t in tiles])
            self.shard_index = index
            self.seq_lens = list(lens)
        else:
            names, seqs = readFasta(path)
            (c0, c1), tiles = plan_tiles([len(s) for s in seqs], w, inc, scaffolds_all, self.kmax, rank, world)
            self.load([seqs[t["scaf"]][t["base0"]:t["end"]] for t in tiles])
            self.seq_lens = [len(s) for s in seqs]
        self.tiles, self.tile_geom = tiles, (w, inc, scaffolds_all)
        return names, (c0, c1)

    @property
    def padded_len(self):
        return self._padded

    def profile_reset(self):
        self.raw[:] = 0
        self.profile = None

    def profile_add(self, mask_host=False, pos_begin=-1, pos_end=-1):
        if self.tiles is not None:                       # a tiled batch: the positions this rank owns
            assert pos_begin < 0 and pos_end < 0
            ranges = [(t["own0"] - t["base0"], t["own1"] - t["base0"]) for t in self.tiles]
            self.raw += N.raw_profile(self.seqs, self.kmin, self.kmax, mask_host, ranges)
            return
        if pos_begin < 0 and pos_end < 0:
            pos_begin, pos_end = 0, self._padded
        ranges = [(pos_begin - o, pos_end - o) for o in self.off]
        self.raw += N.raw_profile(self.seqs, self.kmin, self.kmax, mask_host, ranges)

    def profile_raw(self):
        return self.raw.copy()

    def profile_set_raw(self, raw):
        self.raw = np.asarray(raw, dtype=np.int64).copy()

    def profile_allreduce(self, group=None):
        import torch.distributed as dist
        if dist.is_initialized() and dist.get_world_size(group) > 1:
            self.profile_set_raw(allreduce_raw_host(self.raw, group))

    def profile_finalize(self):
        self.profile = N.finalize_raw(self.raw, self.kmin, self.kmax)

    def profile_get(self):
        sym, (tl, ex, nn) = self.profile
        return sym, tl, ex, nn

    def _candidates(self, w, inc, scaffolds_all):
        out = []
        if self.tiles is not None:
            assert (w, inc, scaffolds_all) == self.tile_geom
            for ti, t in enumerate(self.tiles):          # the tile's windows, in the scaffold's own coordinates
                wins = list(N.iter_windows(t["size"], w, inc, scaffolds_all))[t["j0"]:t["j0"] + t["ncand"]]
                for a, b, start, stop in wins:
                    out.append((ti, a - t["base0"], b - t["base0"], start, stop))
            return out
        for si, e in enumerate(self.seqs):
            for a, b, start, stop in N.iter_windows(e.n, w, inc, scaffolds_all):
                out.append((si, a, b, start, stop))
        return out

    def scan_plan(self, w, inc, scaffolds_all=False):
        return len(self._candidates(w, inc, scaffolds_all))

    def scan(self, w, inc, rip=False, scaffolds_all=False, c0=0, c1=-1, debug=False):
        cands = self._candidates(w, inc, scaffolds_all)
        if c1 < 0:
            c1 = len(cands)
        sym, meta = self.profile
        ig = N.genome_ivom_table(sym, meta, self.kmin, self.kmax)
        n = c1 - c0
        nan = float("nan")
        r = ScanResult(seq_index=np.zeros(n, np.int32), start=np.zeros(n, np.int64), stop=np.zeros(n, np.int64),
                       status=np.zeros(n, np.uint32), kld=np.full(n, nan), gc=np.full(n, nan),
                       pi=np.full(n, nan) if rip else None, si=np.full(n, nan) if rip else None,
                       cri=np.full(n, nan) if rip else None, counts=None, meta=None)
        for t, (si, a, b, start, stop) in enumerate(cands[c0:c1]):
            win = self.seqs[si].slice(a, b)
            r.seq_index[t], r.start[t], r.stop[t] = (self.tiles[si]["scaf"] if self.tiles is not None else si), start, stop
            if (win.n - int(win.upper.sum())) >= 0.3 * win.n:
                continue
            row = N.score_window(win, ig, self.kmin, self.kmax, rip)
            st = _ffi.ROW_KEPT
            if "error" in row:
                st |= _ffi.ROW_ZERO_WEIGHT
            else:
                r.kld[t] = row["KLD"]
            r.status[t] = st
            r.gc[t] = row["GC"]
            if rip:
                r.pi[t], r.si[t], r.cri[t] = row["RIP"]
        r.n_candidates = len(cands)
        return r

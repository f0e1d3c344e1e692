"""Multi-GPU orchestration of the hot path: one process per GPU, torch.distributed (RCCL) for the one
exchange step.

The path shards embarrassingly:
  * phase A - every rank counts the k-mers that start in ITS part of the genome (whole scaffolds, or a
    contiguous padded-position range of one huge scaffold); the raw profile is linear, so ONE
    all-reduce(sum, int64, sum 4^k + 4 elements = 699 KB at k=1..8) gives every rank the genome profile;
  * phase B - windows are independent: a rank scans its scaffolds (or its range of candidate windows);
    rows are gathered to rank 0 in output order.  No other collective.

`run_sharded` works with any object that has the Engine interface, which lets the CPU test-suite rehearse
the N>1 path over gloo with an oracle-backed stand-in.
"""
import numpy as np


def lpt_shards(lens, world):
    """Longest-processing-time bin packing of scaffold lengths onto `world` ranks.
    Returns a list (per rank) of scaffold indices, each in ascending (= output) order."""
    order = sorted(range(len(lens)), key=lambda s: (-lens[s], s))
    load = [0] * world
    bins = [[] for _ in range(world)]
    for s in order:
        r = min(range(world), key=lambda q: (load[q], q))
        bins[r].append(s)
        load[r] += lens[s]
    return [sorted(b) for b in bins]


def split_range(n, rank, world):
    """[lo, hi) of the rank-th of `world` near-equal contiguous parts of range(n)."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def choose_mode(lens, world):
    """'scaffold' when scaffolds balance across ranks within 25 %, else 'range' (split windows / positions
    of the whole batch, e.g. one chromosome on 8 GPUs)."""
    if world == 1 or not lens:
        return "scaffold"
    loads = [sum(lens[s] for s in b) for b in lpt_shards(lens, world)]
    mean = sum(loads) / float(world)
    return "scaffold" if mean > 0 and max(loads) <= 1.25 * mean else "range"


def _dist():
    import torch.distributed as dist
    return dist


def allreduce_raw_host(raw, group=None):
    """Sum a host copy of the raw profile over all ranks (gloo, or any backend that reduces CPU tensors)."""
    import torch
    dist = _dist()
    t = torch.from_numpy(np.ascontiguousarray(raw, dtype=np.int64).copy())
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.numpy()


def run_sharded(engine, names, seqs, w, inc, mask_host=False, rip=False, scaffolds_all=False, mode=None,
                group=None, query=None):
    """Phase A + phase B of one job on this rank; returns the job's rows on rank 0 (None elsewhere).

    names/seqs: the host genome (all ranks read the same FASTA).  query = (names, seqs) if -Q differs.
    Rows: (name, start, stop, status, kld, gc[, pi, si, cri]) in the reference's output order."""
    dist = _dist()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lens = [len(s) for s in seqs]
    mode = mode or choose_mode(lens, world)
    qnames, qseqs = query if query is not None else (names, seqs)
    qlens = [len(s) for s in qseqs]

    # ---- phase A
    engine.profile_reset()
    if mode == "scaffold":
        mine = lpt_shards(lens, world)[rank]
        engine.load([seqs[s] for s in mine])
        engine.profile_add(mask_host=mask_host)
    else:
        engine.load(seqs)
        p0, p1 = split_range(engine.padded_len, rank, world)
        engine.profile_add(mask_host=mask_host, pos_begin=p0, pos_end=p1)
    engine.profile_allreduce(group)
    engine.profile_finalize()

    # ---- phase B
    if mode == "scaffold":
        qmine = lpt_shards(qlens, world)[rank]
        if query is not None:                       # otherwise the rank's host scaffolds are already resident
            engine.load([qseqs[s] for s in qmine])
        res = engine.scan(w, inc, rip=rip, scaffolds_all=scaffolds_all)
        seq_global = np.asarray(qmine, dtype=np.int64)[res.seq_index] if len(res) else np.zeros(0, np.int64)
        cand_key = np.arange(len(res), dtype=np.int64)
    else:
        if query is not None:
            engine.load(qseqs)
        total = engine.scan_plan(w, inc, scaffolds_all)
        c0, c1 = split_range(total, rank, world)
        res = engine.scan(w, inc, rip=rip, scaffolds_all=scaffolds_all, c0=c0, c1=c1)
        seq_global = res.seq_index.astype(np.int64)
        cand_key = np.arange(c0, c1, dtype=np.int64)

    return _gather_rows(dist, group, rank, world, res, seq_global, cand_key, qnames, rip)


def _gather_columns(dist, group, rank, world, local):
    """Per-rank dicts of equally long numpy columns -> list of those dicts on rank 0 (None elsewhere), as TENSORS: one
    all-gather of the row counts, then every rank sends its columns to rank 0 as one byte buffer (point to point: ranks hold
    different numbers of rows).  No pickling of Python objects; over RCCL the bytes travel device to device."""
    import torch
    if world == 1:
        return [local]
    keys = list(local)
    dtypes = [local[k].dtype for k in keys]
    device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    n = len(local[keys[0]]) if keys else 0
    counts = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([n], dtype=torch.int64, device=device), group=group)
    counts = [int(c.item()) for c in counts]
    width = sum(dt.itemsize for dt in dtypes)
    if rank != 0:
        if n:
            buf = np.concatenate([np.ascontiguousarray(local[k]).view(np.uint8).reshape(-1) for k in keys])
            dist.send(torch.from_numpy(buf).to(device), dst=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return None
    parts = [local]
    for r in range(1, world):
        m = counts[r]
        got = {}
        if m:
            t = torch.empty(m * width, dtype=torch.uint8, device=device)
            dist.recv(t, src=dist.get_global_rank(group, r) if group is not None else r, group=group)
            raw = t.cpu().numpy()
            o = 0
            for k, dt in zip(keys, dtypes):
                got[k] = raw[o:o + m * dt.itemsize].view(dt).copy()
                o += m * dt.itemsize
        else:
            got = {k: np.zeros(0, dt) for k, dt in zip(keys, dtypes)}
        parts.append(got)
    return parts


def _gather_rows(dist, group, rank, world, res, seq_global, cand_key, qnames, rip):
    """Rows of all ranks on rank 0, in the reference's output order (scaffold order, then candidate order)."""
    keep = res.kept
    local = {"seq": seq_global[keep], "key": cand_key[keep], "start": res.start[keep], "stop": res.stop[keep],
             "status": res.status[keep], "kld": res.kld[keep], "gc": res.gc[keep]}
    if rip:
        local.update(pi=res.pi[keep], si=res.si[keep], cri=res.cri[keep])

    parts = _gather_columns(dist, group, rank, world, local)
    if parts is None:
        return None
    merged = {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}
    order = np.lexsort((merged["key"], merged["seq"]))          # scaffold order, then candidate order
    rows = []
    for r in order.tolist():
        row = (qnames[int(merged["seq"][r])], int(merged["start"][r]), int(merged["stop"][r]), int(merged["status"][r]),
               float(merged["kld"][r]), float(merged["gc"][r]))
        if rip:
            row += (float(merged["pi"][r]), float(merged["si"][r]), float(merged["cri"][r]))
        rows.append(row)
    return rows


# ---- window-tile sharding with halo: the host-side specification of frisk_fasta_load_shard ----------------
def plan_scaffold(size, w, inc, scaffolds_all):
    """(number of candidate windows, kind) of one scaffold - crawlGenome L194-251: kind 1 = the whole scaffold as one window
    (small scaffold, only with --scaffoldsAll), kind 0 = floor(size / inc) regular windows."""
    if float(size) <= float(w) + ((float(w) * 0.75) - float(inc)):
        return (1 if scaffolds_all else 0), 1
    return (len(range(0, size - inc + 1, inc)) if size - inc + 1 > 0 else 0), 0


def plan_tiles(lens, w, inc, scaffolds_all, kmax, rank, world):
    """What rank `rank` of `world` keeps resident.  Returns ((cand_begin, cand_end), tiles); a tile is a dict
    scaf / size / base0 (first resident base) / end (one past the last) / j0, ncand (its windows inside the scaffold) /
    kind / own0, own1 (the positions whose k-mers this rank counts).  Properties (tested): the candidate ranges of the ranks
    partition the job's numbering, every window's bases are resident on its rank, the owned ranges partition every
    scaffold, and K-1 bases behind every owned range are resident."""
    plans = [plan_scaffold(n, w, inc, scaffolds_all) for n in lens]
    first, total = [], 0
    for nc, _ in plans:
        first.append(total)
        total += nc
    c0, c1 = split_range(total, rank, world)
    tiles = []
    for s, size in enumerate(lens):
        nc, kind = plans[s]
        ja, jb = max(c0, first[s]) - first[s], min(c1, first[s] + nc) - first[s]
        if jb > ja:
            t = dict(scaf=s, size=size, kind=kind, j0=ja, ncand=jb - ja)
            if kind == 1:
                t.update(base0=0, own0=0, own1=size, end=size)
            else:
                a, b = ja * inc, (jb - 1) * inc + w
                if b > size:                                    # the range includes jumpback windows (L230-243)
                    b, a = size, min(a, max(0, size - w))
                own1 = size if jb == nc else jb * inc
                t.update(base0=a, own0=ja * inc, own1=own1, end=max(b, min(size, own1 + kmax - 1)))
            tiles.append(t)
        elif nc == 0:
            mine = (c0 <= first[s] < c1) or (first[s] == total and rank == world - 1) or (total == 0 and rank == world - 1)
            if mine:
                tiles.append(dict(scaf=s, size=size, kind=kind, j0=0, ncand=0, base0=0, own0=0, own1=size, end=size))
    return (c0, c1), tiles


def _gather_table(dist, group, rank, world, res, names, rip):
    """The rows of all ranks on rank 0 as one ScoreTable in the reference's output order.  Ranks hold consecutive candidate
    ranges, so rank order IS output order.  Returns (table, zero_weight) on rank 0 and (None, zero_weight) elsewhere, where
    zero_weight tells every rank whether the reference would have died with ZeroDivisionError (L437)."""
    from . import _ffi
    from .table import ScoreTable
    keep = res.kept
    local = {"seq": res.seq_index[keep], "start": res.start[keep], "stop": res.stop[keep], "status": res.status[keep],
             "kld": res.kld[keep], "gc": res.gc[keep]}
    if rip:
        local.update(pi=res.pi[keep], si=res.si[keep], cri=res.cri[keep])
    parts = _gather_columns(dist, group, rank, world, local)
    table, flag = None, [False]
    if rank == 0:
        m = {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}
        bad = np.nonzero((m["status"] & _ffi.ROW_ZERO_WEIGHT) != 0)[0]
        n = int(bad[0]) if bad.size else len(m["seq"])            # the reference has written every row before the failing one
        flag = [bool(bad.size)]
        int0 = ((m["status"][:n] & _ffi.ROW_NO_MAXMER) != 0).astype(np.uint8)
        table = ScoreTable(names, m["seq"][:n], m["start"][:n], m["stop"][:n], m["kld"][:n], m["gc"][:n],
                           m["pi"][:n] if rip else None, m["si"][:n] if rip else None, m["cri"][:n] if rip else None, int0)
    if world > 1:                               # (one byte, as a tensor)
        import torch
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
        f = torch.tensor([1 if flag[0] else 0], dtype=torch.uint8, device=device)
        dist.broadcast(f, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        flag = [bool(int(f.item()))]
    return table, bool(flag[0])


def check_same_records(engine, names, group=None):
    """Every rank of a sharded job loads its tiles by itself - through a seek index or by parsing (load_fasta_shard) - and
    plans its windows from the record table it ended up with.  Ranks whose tables differ (an index that disagrees with the
    parser, a file that changed between two ranks' reads) would silently duplicate or drop rows: compare a digest of
    (record count, names, lengths) over the ranks with one tiny all-reduce and raise on every rank when it differs."""
    import hashlib
    dist = _dist()
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    import torch
    h = hashlib.sha256()
    lens = [int(x) for x in engine.seq_lens]
    for nm, ln in zip(names, lens):
        h.update(nm.encode("utf-8", "replace") + b"\0" + str(ln).encode() + b"\0")
    d = h.digest()
    sig = [len(lens), sum(lens)] + [int.from_bytes(d[8 * i:8 * i + 8], "little") >> 1 for i in range(3)]
    device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    lo = torch.tensor(sig, dtype=torch.int64, device=device)
    hi = lo.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    if not bool(torch.equal(lo, hi)):
        raise RuntimeError("the ranks of this job read different record tables from the FASTA (%d records, %d bases on rank %d): "
                           "a seek index that disagrees with the file, or a file that changed during the run - "
                           "remove the index (--recalc) and run again" % (sig[0], sig[1], dist.get_rank(group)))


def profile_sharded(engine, host_path, w, inc, mask_host=False, scaffolds_all=False, group=None, index=None):
    """Phase A of one job on this rank (computeKmers genomeMode, L1442): keep the rank's tiles of the host FASTA resident,
    count the k-mers that start in the positions it owns, join the ONE all-reduce, finalise.  Every rank ends with the whole
    genome's profile.  `index`: seek indices to try (fasta.fastaIndexPaths) - with one a rank reads its tiles' bytes only.
    Returns the names of the FASTA's records."""
    dist = _dist()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    names, _ = engine.load_fasta_shard(host_path, w, inc, rank, world, scaffolds_all, index=index)
    check_same_records(engine, names, group)
    engine.profile_reset()
    engine.profile_add(mask_host=mask_host)
    engine.profile_allreduce(group)
    engine.profile_finalize()
    return names


def scan_sharded(engine, query_path, w, inc, rip=False, scaffolds_all=False, group=None, resident_names=None, index=None):
    """Phase B (loop L1478-1494): scan the rank's candidate range of the query FASTA (its tiles are loaded here unless
    `resident_names` says the query's tiles are resident already), gather on rank 0.  Returns (ScoreTable | None, zero_weight)."""
    dist = _dist()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    names = resident_names
    if names is None:
        names, _ = engine.load_fasta_shard(query_path, w, inc, rank, world, scaffolds_all, index=index)
        check_same_records(engine, names, group)
    res = engine.scan(w, inc, rip=rip, scaffolds_all=scaffolds_all)
    return _gather_table(dist, group, rank, world, res, names, rip)


def run_sharded_files(engine, host_path, w, inc, mask_host=False, rip=False, scaffolds_all=False, group=None,
                      query_path=None, index=None, query_index=None):
    """One whole job straight from FASTA files, N ranks: window-tile sharding with halo - every rank parses the file with the
    native reader but keeps resident (and uploads) only the bases of ITS candidate windows and of the positions it counts;
    one all-reduce; rows gathered on rank 0.  Rows as `run_sharded` (rank 0; None elsewhere)."""
    names = profile_sharded(engine, host_path, w, inc, mask_host, scaffolds_all, group, index=index)
    same = query_path is None or query_path == host_path
    table, _zero = scan_sharded(engine, query_path or host_path, w, inc, rip, scaffolds_all, group,
                                resident_names=names if same else None, index=index if same else query_index)
    if table is None:
        return None
    return [r[:3] + (1,) + r[3:] for r in table.rows()]         # (name, start, stop, status, kld, gc[, pi, si, cri])

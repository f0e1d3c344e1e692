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


def _gather_rows(dist, group, rank, world, res, seq_global, cand_key, qnames, rip):
    """Rows of all ranks on rank 0, in the reference's output order (scaffold order, then candidate order)."""
    keep = res.kept
    local = {"seq": seq_global[keep], "key": cand_key[keep], "start": res.start[keep], "stop": res.stop[keep],
             "status": res.status[keep], "kld": res.kld[keep], "gc": res.gc[keep]}
    if rip:
        local.update(pi=res.pi[keep], si=res.si[keep], cri=res.cri[keep])

    if world > 1:
        parts = [None] * world if rank == 0 else None
        dist.gather_object(local, parts, dst=0, group=group)
        if rank != 0:
            return None
    else:
        parts = [local]
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


def run_sharded_files(engine, host_path, w, inc, mask_host=False, rip=False, scaffolds_all=False, group=None,
                      query_path=None):
    """The same job straight from FASTA files: REPLICATED data, sharded work.  Every rank reads the whole file with
    the native reader (`Engine.load_fasta`; a 3 Gb assembly is < 1.3 GB packed, nothing beside 288 GB of HBM), counts
    the k-mers that start in its contiguous range of padded positions, joins the one all-reduce, and scans its
    contiguous range of candidate windows - balanced whatever the scaffold lengths, and no sequence ever exists as
    a Python string.  Rows as `run_sharded`."""
    dist = _dist()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    names = engine.load_fasta(host_path)
    engine.profile_reset()
    p0, p1 = split_range(engine.padded_len, rank, world)
    engine.profile_add(mask_host=mask_host, pos_begin=p0, pos_end=p1)
    engine.profile_allreduce(group)
    engine.profile_finalize()
    qnames = names
    if query_path is not None and query_path != host_path:
        qnames = engine.load_fasta(query_path)
    total = engine.scan_plan(w, inc, scaffolds_all)
    c0, c1 = split_range(total, rank, world)
    res = engine.scan(w, inc, rip=rip, scaffolds_all=scaffolds_all, c0=c0, c1=c1)
    return _gather_rows(dist, group, rank, world, res, res.seq_index.astype(np.int64),
                        np.arange(c0, c1, dtype=np.int64), qnames, rip)

"""The N>1 path on the CPU: two gloo ranks run frisk_amd.distributed.run_sharded with an oracle-backed
engine; the gathered rows and the all-reduced profile must equal the single-process run exactly."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
for p in (REPO, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

from golden_util import INPUTS  # noqa: E402


def _job(mode):
    from oracle import frisk_oracle as O
    if mode == "scaffold":
        recs = list(O.iter_fasta(os.path.join(INPUTS, "smalls.fa"))) + list(O.iter_fasta(os.path.join(INPUTS, "host.fa")))
        return recs, dict(kmin=1, kmax=4, w=400, inc=150, scaffolds_all=True, rip=True)
    recs = list(O.iter_fasta(os.path.join(INPUTS, "markov_islands.fa")))[:1]
    return recs, dict(kmin=1, kmax=5, w=400, inc=100, scaffolds_all=False, rip=False)


def _run(rank, world, port, mode, out_path):
    import torch.distributed as dist
    from fake_engine import FakeEngine
    from frisk_amd.distributed import run_sharded, run_sharded_files
    if world > 1:
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    if mode == "files":         # the CLI's mode: replicated data from FASTA files, host != query
        eng = FakeEngine(1, 4)
        index = query_index = None
        if world > 1:           # the two ranks read their tiles through seek indices (written once, here by rank 0's process id-free name)
            from frisk_amd.fasta import writeFastaIndex
            index, query_index = out_path + ".host.fai", out_path + ".query.fai"
            if rank == 0:
                assert writeFastaIndex(os.path.join(INPUTS, "host.fa"), index) and writeFastaIndex(os.path.join(INPUTS, "query.fa"), query_index)
            dist.barrier()
        rows = run_sharded_files(eng, os.path.join(INPUTS, "host.fa"), 400, 150, rip=True, scaffolds_all=True,
                                 query_path=os.path.join(INPUTS, "query.fa"), index=index, query_index=query_index)
        assert (eng.shard_index is not None) == (world > 1)
    else:
        recs, kw = _job(mode)
        eng = FakeEngine(kw["kmin"], kw["kmax"])
        rows = run_sharded(eng, [n for n, _ in recs], [s for _, s in recs], kw["w"], kw["inc"], rip=kw["rip"],
                           scaffolds_all=kw["scaffolds_all"], mode=mode)
    sym, tl, ex, nn = eng.profile_get()
    if rank == 0:
        np.save(out_path, np.array([rows, sym, (tl, ex, nn)], dtype=object), allow_pickle=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["scaffold", "range", "files"])
def test_two_ranks_equal_one(tmp_path, mode):
    single = str(tmp_path / "single.npy")
    double = str(tmp_path / "double.npy")
    _run(0, 1, 0, mode, single)
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_run, args=(2, port, mode, double), nprocs=2, join=True)
    a = np.load(single, allow_pickle=True)
    b = np.load(double, allow_pickle=True)
    assert len(a[0]) > 5
    assert np.array_equal(a[1], b[1]) and tuple(a[2]) == tuple(b[2])      # profile: identical after the all-reduce
    assert len(a[0]) == len(b[0])
    for ra, rb in zip(a[0], b[0]):
        assert ra[:4] == rb[:4]
        for x, y in zip(ra[4:], rb[4:]):
            assert (x != x and y != y) or x == y


def test_lpt_and_ranges():
    from frisk_amd.distributed import choose_mode, lpt_shards, split_range
    lens = [230218, 813184, 316620, 1531933, 576874, 270161, 1090940, 562643]
    bins = lpt_shards(lens, 2)
    assert sorted(bins[0] + bins[1]) == list(range(len(lens)))
    loads = [sum(lens[s] for s in b) for b in bins]
    assert max(loads) <= 1.1 * min(loads)
    assert choose_mode(lens, 2) == "scaffold" and choose_mode([248956422], 8) == "range"
    parts = [split_range(1001, r, 8) for r in range(8)]
    assert parts[0][0] == 0 and parts[-1][1] == 1001 and all(parts[i][1] == parts[i + 1][0] for i in range(7))


def test_tile_plan_properties():
    """plan_tiles (the specification of frisk_fasta_load_shard): candidate ranges partition the job's numbering, every
    window's bases are resident on its rank, owned ranges partition every scaffold, K-1 bases of halo follow each."""
    from frisk_amd.distributed import plan_scaffold, plan_tiles
    from oracle import frisk_oracle_np as N
    rng = np.random.default_rng(12)
    for trial in range(60):
        w = int(rng.choice([400, 2000, 5000]))
        inc = int(rng.choice([w // 10, w // 5, w // 2, w, w + w // 4, 3 * w]))
        all_ = bool(rng.integers(0, 2))
        lens = [int(x) for x in rng.choice([0, 7, w // 2, w, w + 1, 2 * w - 1, 3 * w + 17, 10 * w + 3, 57 * w], size=int(rng.integers(1, 9)))]
        world = int(rng.choice([1, 2, 3, 5, 8]))
        kmax = 8
        total = sum(plan_scaffold(n, w, inc, all_)[0] for n in lens)
        owned = [np.zeros(n, np.int32) for n in lens]
        covered = []
        prev_end = 0
        for rank in range(world):
            (c0, c1), tiles = plan_tiles(lens, w, inc, all_, kmax, rank, world)
            assert c0 == prev_end and c1 >= c0
            prev_end = c1
            nloc = 0
            for t in tiles:
                s, size = t["scaf"], t["size"]
                assert size == lens[s] and 0 <= t["base0"] <= t["own0"] <= t["own1"] <= t["end"] <= size
                owned[s][t["own0"]:t["own1"]] += 1
                assert t["end"] >= min(size, t["own1"] + kmax - 1)
                wins = list(N.iter_windows(size, w, inc, all_))[t["j0"]:t["j0"] + t["ncand"]]
                assert len(wins) == t["ncand"]
                for a, b, _, _ in wins:
                    assert t["base0"] <= a and b <= t["end"]
                nloc += t["ncand"]
            assert nloc == c1 - c0
            covered.append(nloc)
        assert prev_end == total and sum(covered) == total
        for o in owned:
            assert np.all(o == 1)


def _run_mismatch(rank, world, port, out_path):
    import torch.distributed as dist
    from fake_engine import FakeEngine
    from frisk_amd.distributed import check_same_records
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    eng = FakeEngine(1, 4)
    names = ["a", "b"]
    eng.seq_lens = [100, 200]
    check_same_records(eng, names)                          # the same table everywhere: silent
    eng.seq_lens = [100, 200 + rank]                        # rank 1 read another length for record b
    try:
        check_same_records(eng, names)
        verdict = "silent"
    except RuntimeError as err:
        verdict = "raised: " + str(err)[:60]
    with open("%s.%d" % (out_path, rank), "w") as fh:
        fh.write(verdict)
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_with_different_record_tables_fail_loudly(tmp_path):
    """every rank decides by itself whether it loads its tiles through an index or by parsing; if the record tables
    they end with differ, window planning differs and rows would be duplicated or lost - the job must stop on EVERY rank."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = str(tmp_path / "verdict")
    mp.spawn(_run_mismatch, args=(2, port, out), nprocs=2, join=True)
    for r in range(2):
        assert open("%s.%d" % (out, r)).read().startswith("raised: the ranks of this job read different record tables")

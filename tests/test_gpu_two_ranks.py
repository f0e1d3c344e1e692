"""The N > 1 product path on REAL kernels: two ranks on the one GPU of the test box (gloo for the exchange - RCCL refuses two ranks on
one device), each with its own Engine, run `frisk_amd.distributed.run_sharded_files`: window tiles + halo from the FASTA file
(frisk_fasta_load_shard_indexed: each rank copies the bytes of its tiles from the mapped file), the raw profiles summed by the all-reduce, every rank scanning its candidate range with the chunked /
sliding kernels, rows gathered on rank 0 as tensors.  Rows and profile must equal the one-rank run bit for bit."""
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

pytestmark = pytest.mark.gpu

GEOM = dict(kmin=1, kmax=8, w=5000, inc=1000)


def _write_fasta(path):
    from frisk_amd import synth
    lens = [1_300_000, 64_000, 7_001, 0, 250_000, 9_999]
    with open(path, "wb") as fh:
        for i, n in enumerate(lens):
            s = synth.scaffold(n, 77, i, island_frac=0.1, n_frac=0.06, lower_frac=0.03, repeats_per_kb=0.1)
            fh.write(b">scaf%d two ranks\n" % i)
            for o in range(0, len(s), 80):
                fh.write(s[o:o + 80] + b"\n")


def _run(rank, world, port, fasta, out_path, index=None):
    import torch.distributed as dist
    from frisk_amd.distributed import run_sharded_files
    from frisk_amd.engine import Engine
    if world > 1:
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    with Engine(GEOM["kmin"], GEOM["kmax"], device=0) as eng:
        rows = run_sharded_files(eng, fasta, GEOM["w"], GEOM["inc"], rip=True, scaffolds_all=True, index=index)
        assert (eng.shard_index is not None) == (index is not None)
        sym, tl, ex, nn = eng.profile_get()
        resident = eng.padded_len
    if rank == 0:
        np.save(out_path, np.array([rows, sym, (tl, ex, nn), resident], dtype=object), allow_pickle=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_two_real_ranks_equal_one(tmp_path):
    fasta = str(tmp_path / "genome.fa")
    _write_fasta(fasta)
    single, double = str(tmp_path / "single.npy"), str(tmp_path / "double.npy")
    port = 29700 + (os.getpid() % 2000)
    mp.spawn(_run, args=(1, port, fasta, single), nprocs=1, join=True)
    # (the two ranks read their tiles through the seek index, the single rank parses the file)
    from frisk_amd.fasta import writeFastaIndex
    assert writeFastaIndex(fasta, fasta + ".frisk.fai") == 6
    mp.spawn(_run, args=(2, port + 1, fasta, double, fasta + ".frisk.fai"), nprocs=2, join=True)
    a = np.load(single, allow_pickle=True)
    b = np.load(double, allow_pickle=True)
    assert len(a[0]) > 1500
    assert np.array_equal(np.asarray(a[1]), np.asarray(b[1])) and tuple(a[2]) == tuple(b[2])        # the all-reduced profile
    assert len(a[0]) == len(b[0])
    for ra, rb in zip(a[0], b[0]):                                                                   # rows, bit for bit (NaN == NaN)
        assert ra[:4] == rb[:4]
        assert np.array_equal(np.array(ra[4:], dtype=np.float64).view(np.uint64), np.array(rb[4:], dtype=np.float64).view(np.uint64))
    assert b[3] < 0.75 * a[3]                                                                        # rank 0 held its tiles, not the genome

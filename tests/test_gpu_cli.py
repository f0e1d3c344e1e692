"""End-to-end CLI on the GPU: `python -m frisk_amd` semantics - score table text, caches, GFF3."""
import json
import os

import pytest

from golden_util import GOLD, INPUTS, Case

pytestmark = pytest.mark.gpu

WRITERS = json.load(open(os.path.join(GOLD, "writers.json")))


def test_cli_table_caches_and_gff(tmp_path, capsys, monkeypatch):
    from frisk_amd import postprocess as pp
    from frisk_amd.cli import main
    c = Case("markov_k6")
    out = tmp_path / "T"
    g = WRITERS["e2e"]["markov_k6"]      # what the reference's own thresholdKLD / anomaly2GFF / thresholdRIP / RIP2GFF write
    argv = ["-H", c.host, "-k", "6", "-w", "400", "-i", "150", "--RIP", "-t", str(out), "-F", str(g["forceThresholdKLD"]),
            "--gffOutfile", "anom.gff3", "--mergeDist", str(g["mergeDist"])]
    for opt, val in g["rip_args"].items():
        argv += ["--" + opt + "=" + str(val)]
    assert main(argv) == 0
    printed = capsys.readouterr().out.splitlines()
    table = open(out / "raw_window_scores.bed").read().splitlines()
    assert table[0] == "name\tstart\tstop\twindowKLD\tGC\tPI\tSI\tCRI"
    assert len(table) == 1 + len(c.rows)
    assert printed[0] == "frisk -- 0+unknown" and printed[1:] == table[1:]              # every row echoed (L1494)
    same_text = 0
    for line, exp in zip(table[1:], c.rows):
        f = line.split("\t")
        assert f[:3] == [exp["name"], str(exp["start"]), str(exp["stop"])]
        # 12 significant digits, as py2 prints; a KLD that differs from the reference's by ~1e-14 can land on the
        # other side of a rounding boundary of the 12th digit, so the text is compared numerically and counted
        assert abs(float(f[3]) - exp["KLD"]) <= 1e-11 and len(f[3]) <= 16
        same_text += f[3] == pp.py2_str(exp["KLD"])
        assert f[4] == pp.py2_str(exp["GC"])
        assert f[5:] == [pp.py2_str(v) for v in exp["RIP"]]
    assert same_text >= 0.9 * len(c.rows)
    assert os.path.isfile(out / c.doc["genome_pickle_basename"]) and os.path.isfile(out / c.doc["window_pickle_basename"])
    # both feature files byte for byte: the reference's post-processing on the reference's scores for these windows
    assert open(out / "anom.gff3").read() == g["anomaly_gff"] and g["anomaly_gff"].count("\n") > 3
    assert open(out / "RIP_annotation.gff3").read() == g["rip_gff"] and g["rip_gff"].count("\n") > 1
    # second run: both caches are reused (no recomputation), same table; py3 float text on request
    monkeypatch.setenv("FRISK_FLOAT_REPR", "py3")
    os.remove(out / "raw_window_scores.bed")
    assert main(argv) == 0
    assert not os.path.exists(out / "raw_window_scores.bed")                           # table only written when scores are computed
    assert main(argv + ["--recalcWin", "--exitAfter", "WindowKLD"]) == 0
    table3 = open(out / "raw_window_scores.bed").read().splitlines()
    assert len(table3[1].split("\t")[3]) >= 16                                          # repr(), not 12 digits
    assert abs(float(table3[1].split("\t")[3]) - c.rows[0]["KLD"]) < 1e-12


def test_cli_zero_weight_raises_like_reference(tmp_path):
    from frisk_amd.cli import main
    c = Case("hq_m5k6_zero")
    with pytest.raises(ZeroDivisionError):
        main(["-H", c.host, "-Q", c.query, "-m", "5", "-k", "6", "-w", "500", "-i", "100", "-t", str(tmp_path / "T"),
              "--exitAfter", "WindowKLD"])


def test_cli_sharded_path_in_one_rank_group(tmp_path, capsys, monkeypatch):
    """The multi-GPU code path of the CLI (run_sharded + RCCL all-reduce + row gather) rehearsed in a one-rank nccl
    group on a single GPU: same table as the plain path."""
    import torch.distributed as dist
    from frisk_amd.cli import main
    c = Case("smalls_all")
    argv = ["-H", c.host, "-k", "4", "-w", "400", "-i", "150", "--RIP", "--scaffoldsAll", "--exitAfter", "WindowKLD"]
    assert main(argv + ["-t", str(tmp_path / "A")]) == 0
    plain = open(tmp_path / "A" / "raw_window_scores.bed").read()
    for k, v in (("FRISK_FORCE_SHARDED", "1"), ("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0"),
                 ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29641")):
        monkeypatch.setenv(k, v)
    try:
        assert main(argv + ["-t", str(tmp_path / "B")]) == 0
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()
    assert open(tmp_path / "B" / "raw_window_scores.bed").read() == plain
    assert len(plain.splitlines()) == 1 + len(c.rows)


@pytest.mark.parametrize("case", ["markov_m2k4", "k8_w2000", "markov_k6"])
def test_cli_hmm_segmentation_gff(tmp_path, case):
    """--hmmKLD --hmmOutfile end to end on the GPU (reference L1537-1548): the GFF3 files equal, byte for byte, what the
    REFERENCE'S hmm2BED L757-785 / findBaseRanges / range2interval / hmmBED2GFF L589-596 and thresholdKLD / anomaly2GFF
    write for the reference's scores of these windows (tests/golden/writers.json).  Only the model's numbers are this
    package's own (hmmlearn is absent: the golden was made by handing this package's model to the reference's hmm2BED);
    the GPU's scores, within 1e-11 of the reference's, move no window across a state or threshold boundary."""
    from frisk_amd.cli import main
    c = Case(case)
    g = WRITERS["e2e"][case]
    a = c.doc["args"]
    out = tmp_path / "H"
    argv = ["-H", c.host, "-m", str(a["minWordSize"]), "-k", str(a["maxWordSize"]), "-w", str(a["windowlen"]),
            "-i", str(a["increment"]), "-t", str(out), "-F", str(g["forceThresholdKLD"]), "--mergeDist", str(g["mergeDist"]),
            "--hmmKLD", "--hmmOutfile", "states.gff3", "--gffOutfile", "anom.gff3"]
    if g["rip_args"] is not None:
        argv += ["--RIP"] + ["--" + opt + "=" + str(val) for opt, val in g["rip_args"].items()]
    assert main(argv) == 0
    assert len(open(out / "raw_window_scores.bed").read().splitlines()) == 1 + len(c.rows)
    assert open(out / "states.gff3").read() == g["hmm_gff"]
    assert open(out / "anom.gff3").read() == g["anomaly_gff"]
    if g["rip_args"] is not None:
        assert open(out / "RIP_annotation.gff3").read() == g["rip_gff"]
    if case != "markov_k6":
        assert "State1" in g["hmm_gff"] and "State2" in g["hmm_gff"]


def test_cli_sharded_cache_semantics(tmp_path, monkeypatch):
    """Under torchrun the CLI keeps the reference's cache semantics (L1436-1447, L1454-1459, L1503-1505): --exitAfter
    GenomeKmers stops BEFORE any window is scored and leaves the genome pickle (the one file that option exists to
    produce), equal to the plain run's; a second run reuses both caches; --recalc / --recalcWin force recomputation."""
    import pickle
    import torch.distributed as dist
    from frisk_amd.cli import main
    c = Case("markov_k6")
    base = ["-H", c.host, "-k", "6", "-w", "400", "-i", "150", "--RIP"]
    assert main(base + ["-t", str(tmp_path / "P"), "--exitAfter", "WindowKLD"]) == 0
    plain_gen = pickle.load(open(tmp_path / "P" / c.doc["genome_pickle_basename"], "rb"))
    plain_tab = open(tmp_path / "P" / "raw_window_scores.bed").read()
    for k, v in (("FRISK_FORCE_SHARDED", "1"), ("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0"),
                 ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29643")):
        monkeypatch.setenv(k, v)
    S = tmp_path / "S"
    try:
        assert main(base + ["-t", str(S), "--exitAfter", "GenomeKmers"]) == 0
        assert os.path.isfile(S / c.doc["genome_pickle_basename"])
        assert not os.path.exists(S / "raw_window_scores.bed") and not os.path.exists(S / c.doc["window_pickle_basename"])
        assert pickle.load(open(S / c.doc["genome_pickle_basename"], "rb")) == plain_gen
        t_gen = os.path.getmtime(S / c.doc["genome_pickle_basename"])
        assert main(base + ["-t", str(S), "--exitAfter", "WindowKLD"]) == 0             # profile from the cache, scan sharded
        assert os.path.getmtime(S / c.doc["genome_pickle_basename"]) == t_gen
        assert open(S / "raw_window_scores.bed").read() == plain_tab
        frame = pickle.load(open(S / c.doc["window_pickle_basename"], "rb"))             # a DataFrame, as the reference pickles
        assert list(frame.columns) == ["name", "start", "stop", "windowKLD", "GC", "PI", "SI", "CRI"] and len(frame) == len(c.rows)
        os.remove(S / "raw_window_scores.bed")
        assert main(base + ["-t", str(S), "-F", "0.08", "--gffOutfile", "a.gff3"]) == 0  # both caches: nothing recomputed
        assert not os.path.exists(S / "raw_window_scores.bed") and os.path.isfile(S / "a.gff3")
        assert main(base + ["-t", str(S), "--recalc", "--recalcWin", "--exitAfter", "WindowKLD"]) == 0
        assert os.path.getmtime(S / c.doc["genome_pickle_basename"]) > t_gen
        assert open(S / "raw_window_scores.bed").read() == plain_tab
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_cli_sharded_seek_index_life_cycle(tmp_path, monkeypatch, caplog):
    """A sharded job that had to parse the FASTA leaves a seek index beside the other caches (rank 0, background thread); the next
    one reads its tiles through it - same table; --recalc ignores and rewrites it; an index made from another version of the file
    is not used."""
    import logging
    import torch.distributed as dist
    from frisk_amd.cli import main
    c = Case("markov_k6")
    fasta = tmp_path / "genome.fa"
    fasta.write_bytes(open(c.host, "rb").read())
    base = ["-H", str(fasta), "-k", "6", "-w", "400", "-i", "150", "--RIP", "--exitAfter", "WindowKLD", "--recalcWin"]
    assert main(base + ["-t", str(tmp_path / "P")]) == 0
    plain = open(tmp_path / "P" / "raw_window_scores.bed").read()
    assert not os.path.exists(tmp_path / "P" / "genome.fa.frisk.fai")            # (one rank: nothing to shard, no index)
    for k, v in (("FRISK_FORCE_SHARDED", "1"), ("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0"),
                 ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29647")):
        monkeypatch.setenv(k, v)
    S = tmp_path / "S"
    idx = S / "genome.fa.frisk.fai"
    genome_pickle = S / c.doc["genome_pickle_basename"].replace(os.path.basename(c.host), "genome.fa")
    try:
        with caplog.at_level(logging.INFO):
            assert main(base + ["-t", str(S)]) == 0
        assert "through the index" not in caplog.text and os.path.isfile(idx)
        assert open(S / "raw_window_scores.bed").read() == plain
        stamp = open(idx).readline()
        caplog.clear()
        os.remove(genome_pickle)                          # (so that phase A runs, and loads, again)
        with caplog.at_level(logging.INFO):
            assert main(base + ["-t", str(S)]) == 0
        assert "through the index %s" % idx in caplog.text
        assert open(S / "raw_window_scores.bed").read() == plain
        caplog.clear()
        t_idx = os.stat(idx).st_mtime_ns
        with caplog.at_level(logging.INFO):
            assert main(base + ["-t", str(S), "--recalc"]) == 0                   # recompute: parse, and write the index again
        assert "through the index" not in caplog.text and os.stat(idx).st_mtime_ns > t_idx
        # the FASTA changes: the old index is refused (stamp), the job parses and replaces it
        data = fasta.read_bytes()
        fasta.write_bytes(data[:40] + data[40:].replace(b"A", b"C", 7))
        os.remove(genome_pickle)
        caplog.clear()
        with caplog.at_level(logging.INFO):
            assert main(base + ["-t", str(S)]) == 0
        assert "through the index" not in caplog.text and open(idx).readline() != stamp
        assert open(S / "raw_window_scores.bed").read() != plain
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_cli_zero_weight_writes_the_rows_before_the_failure(tmp_path, capsys):
    """The reference writes and prints each row before scoring the next (L1487-1494): when a window dies with
    ZeroDivisionError (L437) the table already holds every row before it."""
    from frisk_amd.cli import main
    c = Case("hq_m5k6_zero")
    first_bad = next(i for i, r in enumerate(c.rows) if "error" in r)
    with pytest.raises(ZeroDivisionError):
        main(["-H", c.host, "-Q", c.query, "-m", "5", "-k", "6", "-w", "500", "-i", "100", "-t", str(tmp_path / "Z"),
              "--exitAfter", "WindowKLD"])
    table = open(tmp_path / "Z" / "raw_window_scores.bed").read().splitlines()
    assert len(table) == 1 + first_bad
    assert capsys.readouterr().out.splitlines()[1:] == table[1:]
    assert not any(f.endswith("KLD_window_500_increment_100.p") for f in os.listdir(tmp_path / "Z"))


@pytest.mark.parametrize("case", ["smalls_all", "smalls_skip", "nheavy_k4", "overshoot"])
def test_cli_logs_the_reference_progress_lines(tmp_path, caplog, case):
    """crawlGenome's per-scaffold lines (L212-250) from the GPU scan's own candidate arrays equal the reference's."""
    import logging
    from frisk_amd.cli import main
    c = Case(case)
    a = c.doc["args"]
    argv = ["-H", c.host, "-m", str(a["minWordSize"]), "-k", str(a["maxWordSize"]), "-w", str(a["windowlen"]),
            "-i", str(a["increment"]), "-t", str(tmp_path / "L"), "--exitAfter", "WindowKLD"]
    argv += ["--scaffoldsAll"] if a["scaffoldsAll"] else []
    with caplog.at_level(logging.INFO, logger="frisk_amd"):
        assert main(argv) == 0
    want = [ln for ln in WRITERS["crawl_log"][case] if not ln.startswith("Window from ")]
    got = [r.getMessage() for r in caplog.records if r.name == "frisk_amd"]
    first = got.index(want[0])
    assert got[first:first + len(want)] == want


def test_cli_packed_sequence_cache(tmp_path, capsys):
    """<fasta>.frisk2bit beside the pickle caches: written by the first run, used by the second (no parse, no pack: the table is
    the same text), ignored and rewritten with --recalc, and not trusted once the FASTA has changed."""
    import shutil
    from frisk_amd.cli import main
    from frisk_amd.hotpath import readSeqCache
    c = Case("k8_w2000")
    fa = tmp_path / "genome.fa"
    shutil.copy(c.host, fa)
    T = tmp_path / "T"
    argv = ["-H", str(fa), "-k", "8", "-w", "2000", "-i", "500", "-t", str(T), "--recalcWin", "--exitAfter", "WindowKLD"]
    assert main(argv) == 0
    first = open(T / "raw_window_scores.bed").read()
    cache = T / "genome.fa.frisk2bit"
    assert cache.is_file() and readSeqCache(str(cache), str(fa)) is not None
    assert len(first.splitlines()) == 1 + len(c.rows)
    t0 = os.path.getmtime(cache)
    assert main(argv) == 0                                          # the profile pickle and the sequence cache are both reused
    assert open(T / "raw_window_scores.bed").read() == first and os.path.getmtime(cache) == t0
    assert main(argv + ["--recalc"]) == 0                           # --recalc: recompute - and rewrite the cache
    assert open(T / "raw_window_scores.bed").read() == first and os.path.getmtime(cache) >= t0
    # the FASTA changes (one more record): the cache is for another file now
    with open(fa, "a") as fh:
        fh.write(">extra\n" + "ACGGTCA" * 700 + "\n")         # 4 900 bases: long enough for windows of its own
    assert readSeqCache(str(cache), str(fa)) is None
    assert main(argv + ["--recalc"]) == 0
    assert len(open(T / "raw_window_scores.bed").read().splitlines()) > len(first.splitlines())
    capsys.readouterr()


@pytest.mark.parametrize("world", [2, 3])
def test_cli_under_torchrun_ranks_on_one_gpu(tmp_path, world):
    """`torchrun --nproc-per-node N -m frisk_amd` as N real processes (FRISK_DIST_REHEARSAL=1: every rank on device 0, gloo in
    RCCL's place - RCCL refuses two ranks on one device): each rank loads ITS window tiles, the raw profiles are all-reduced,
    every rank scans its tiles and rank 0 gathers the rows - the score table, the window cache's rows and both GFF3 files equal
    the one-process run's byte for byte, on a multi-scaffold genome with invalid runs, lowercase and scaffolds below a window."""
    import pickle
    import subprocess
    import sys
    import numpy as np
    from frisk_amd.cli import main
    rng = np.random.default_rng(20 + world)
    fasta = tmp_path / "genome.fa"
    with open(fasta, "w") as fh:
        for s, n in enumerate([260_000, 90_011, 1_500, 41_003, 7_000, 130_500]):
            seq = rng.choice(np.frombuffer(b"ATGC", dtype=np.uint8), size=n, p=[0.3, 0.3, 0.2, 0.2])
            for _ in range(6):
                a = int(rng.integers(0, n - 10))
                seq[a:a + int(rng.choice([1, 40, 900]))] = ord("N")
                b = int(rng.integers(0, n - 10))
                seq[b:b + 300] |= 0x20
                t = int(rng.integers(0, n - 10))
                seq[t:t + 120] = np.frombuffer((b"CA" * 60), dtype=np.uint8)[:len(seq[t:t + 120])]
            text = seq.tobytes().decode()
            fh.write(">scaf%d some description\n" % s)
            fh.write("\n".join(text[i:i + 70] for i in range(0, n, 70)) + "\n")
    base = ["-H", str(fasta), "-k", "8", "-w", "5000", "-i", "1000", "--RIP", "-F", "0.02", "--gffOutfile", "a.gff3", "--scaffoldsAll"]
    assert main(base + ["-t", str(tmp_path / "P")]) == 0
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(FRISK_DIST_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               PYTHONPATH=os.pathsep.join([os.path.dirname(os.path.dirname(os.path.abspath(__file__)))] + sys.path))
    run = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                          "--master-addr", "127.0.0.1", "--master-port", str(29700 + world), "-m", "frisk_amd"] + base +
                         ["-t", str(tmp_path / "S")], env=env, capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stderr[-3000:]
    plain = open(tmp_path / "P" / "raw_window_scores.bed").read()
    assert plain.count("\n") > 400
    assert open(tmp_path / "S" / "raw_window_scores.bed").read() == plain
    assert open(tmp_path / "S" / "a.gff3").read() == open(tmp_path / "P" / "a.gff3").read()
    rip_p, rip_s = tmp_path / "P" / "RIP_annotation.gff3", tmp_path / "S" / "RIP_annotation.gff3"
    assert os.path.exists(rip_p) == os.path.exists(rip_s) and (not os.path.exists(rip_p) or open(rip_p).read() == open(rip_s).read())
    pick = [f for f in os.listdir(tmp_path / "P") if f.endswith(".p")]
    assert len(pick) == 2
    for f in pick:
        a, b = pickle.load(open(tmp_path / "P" / f, "rb")), pickle.load(open(tmp_path / "S" / f, "rb"))
        assert a.equals(b) if hasattr(a, "equals") else a == b, f
    # (every row echoed once, by rank 0 only: L1494)
    assert run.stdout.count("\n") >= plain.count("\n")

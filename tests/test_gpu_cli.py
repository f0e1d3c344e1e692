"""End-to-end CLI on the GPU: `python -m frisk_amd` semantics - score table text, caches, GFF3."""
import json
import os

import pytest

from golden_util import GOLD, INPUTS, Case

pytestmark = pytest.mark.gpu


def test_cli_table_caches_and_gff(tmp_path, capsys, monkeypatch):
    from frisk_amd import postprocess as pp
    from frisk_amd.cli import main
    c = Case("markov_k6")
    out = tmp_path / "T"
    argv = ["-H", c.host, "-k", "6", "-w", "400", "-i", "150", "--RIP", "-t", str(out), "-F", "0.08", "--gffOutfile", "anom.gff3",
            "--mergeDist", "10"]
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
    gff = open(out / "anom.gff3").read().splitlines()
    assert gff[0] == "##gff-version 3" and all(g.split("\t")[1] == "frisk_0+unknown" for g in gff[1:])
    n_hot = sum(1 for r in c.rows if r["KLD"] >= 0.08)
    assert 1 <= len(gff) - 1 <= n_hot
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


def test_cli_hmm_segmentation_gff(tmp_path):
    """--hmmKLD --hmmOutfile end to end on the GPU (reference L1537-1548, hmm2BED L757-785, hmmBED2GFF L589-596): the
    GFF3 text the CLI writes equals the text that the host-side model produces from the score table the same run wrote
    (the model's numbers are parity-unpinned - hmmlearn is absent - but stacking, per-scaffold decoding, run extraction,
    ordering and the GFF3 layout are the reference's)."""
    from frisk_amd import postprocess as pp
    from frisk_amd.cli import main
    from frisk_amd.hmm import hmm2BED, hmmBED2GFF
    c = Case("markov_k6")
    out = tmp_path / "H"
    assert main(["-H", c.host, "-k", "6", "-w", "400", "-i", "150", "-t", str(out), "-F", "0.08", "--hmmKLD",
                 "--hmmOutfile", "states.gff3", "--gffOutfile", "anom.gff3"]) == 0
    table = [ln.split("\t") for ln in open(out / "raw_window_scores.bed").read().splitlines()[1:]]
    assert len(table) == len(c.rows)
    rows = [(f[0], int(f[1]), int(f[2]), exp["KLD"], exp["GC"]) for f, exp in zip(table, c.rows)]
    gff = open(out / "states.gff3").read()
    lines = gff.splitlines()
    assert lines[0] == "##gff-version 3"
    feats = [ln.split("\t") for ln in lines[1:]]
    assert len(feats) >= 2 and all(len(f) == 9 for f in feats)
    assert all(f[1] == "frisk_" + pp.FRISK_VERSION for f in feats)
    assert {f[0] for f in feats} <= {r[0] for r in rows}
    assert all(1 <= int(f[3]) <= int(f[4]) for f in feats)
    # the same model on the reference's golden KLD values for these windows gives the same features: the GPU's scores
    # (within 1e-11 of the reference's) do not move any window across a state boundary
    intervals, _ = hmm2BED(rows)
    assert "".join(hmmBED2GFF(intervals)) == gff
    assert open(out / "anom.gff3").read().startswith("##gff-version 3")


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

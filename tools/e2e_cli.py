#!/usr/bin/env python3
"""End-to-end timing of `python -m frisk_amd` on synthetic FASTA files of BASELINE.json's shapes (GPU box).
Writes the FASTA (60-column lines, plain and .gz), runs the CLI with --exitAfter WindowKLD (stdout -> /dev/null) and
prints one JSON line per run with the CLI's own timing split (FRISK_TIMING=1).  usage: e2e_cli.py C3|C5shard|C5 [workdir]
E2E_FULL=1 adds the WHOLE pipeline (reference main() L1400-1851 as far as this package goes): scan -> log10 / Otsu threshold ->
2-state HMM segmentation -> merged anomaly features -> RIP features, three GFF3 files
(--RIP --hmmKLD --hmmOutfile ... --threshTypeKLD otsu --gffOutfile ...), cold and from the caches."""
import gzip
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from frisk_amd import Engine, synth  # noqa: E402

shape = sys.argv[1] if len(sys.argv) > 1 else "C3"
work = sys.argv[2] if len(sys.argv) > 2 else "/tmp/frisk_e2e"
os.makedirs(work, exist_ok=True)
if shape == "C3":
    lens, nfrac = synth.C3_LENS, 0.001
elif shape == "C5shard":
    lens, nfrac = synth.c5_shard_lens(8, 0), 0.07
else:
    lens, nfrac = [n for r in range(8) for n in synth.c5_shard_lens(8, r)], 0.07

fa = os.path.join(work, shape + ("_repeats" if os.environ.get("E2E_REPEATS") else "") + ".fa")
t0 = time.time()
if not os.path.exists(fa):
    with Engine(1, 4) as e, open(fa, "wb") as fh:
        # (E2E_REPEATS: simple repeats per kb, frisk_amd/synth.py - 0.35 is the bench's `realistic` shape: the CLI's scan then samples, and takes
        #  the 4-bit form with the side table)
        e.synth(lens, seed=0xE2E, island_frac=0.02, n_frac=nfrac, lower_frac=0.02, repeats_per_kb=float(os.environ.get("E2E_REPEATS", "0")))
        for i, n in enumerate(lens):
            s = np.frombuffer(e.read_seq(i), dtype=np.uint8)
            fh.write(b">scaffold_%d synthetic\n" % i)
            full = (n // 60) * 60
            if full:
                body = np.empty((full // 60, 61), np.uint8)
                body[:, :60] = s[:full].reshape(-1, 60)
                body[:, 60] = 10
                fh.write(body.tobytes())
            if n > full:
                fh.write(s[full:].tobytes() + b"\n")
print(json.dumps({"fasta": fa, "bases": sum(lens), "bytes": os.path.getsize(fa), "write_s": round(time.time() - t0, 2)}), flush=True)
gz = fa + ".gz"
if shape != "C5" and not os.path.exists(gz):
    t0 = time.time()
    with open(fa, "rb") as src, gzip.open(gz, "wb", compresslevel=1) as dst:
        while True:
            b = src.read(1 << 24)
            if not b:
                break
            dst.write(b)
    print(json.dumps({"gz": gz, "bytes": os.path.getsize(gz), "gzip_s": round(time.time() - t0, 2)}), flush=True)

def gff_lines(path):
    return sum(1 for ln in open(path) if not ln.startswith("#")) if os.path.exists(path) else None


if os.environ.get("E2E_FULL"):
    for label, extra in (("full pipeline, cold caches", ["--recalc", "--recalcWin"]),
                         ("full pipeline, sequence cache + genome pickle", ["--recalcWin"]),
                         ("full pipeline, from the window cache", [])):
        tmp = os.path.join(work, "F_" + os.path.basename(fa))
        cmd = [sys.executable, "-m", "frisk_amd", "-H", fa, "-k", "8", "-w", "5000", "-i", "1000", "-t", tmp, "--RIP", "--hmmKLD",
               "--hmmOutfile", "hmm.gff3", "--threshTypeKLD", "otsu", "--gffOutfile", "anomalies.gff3"] + extra
        env = dict(os.environ, FRISK_TIMING="1", PYTHONPATH=ROOT)
        t0 = time.time()
        out = subprocess.run(cmd, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, cwd=ROOT)
        wall = time.time() - t0
        split = None
        for line in out.stderr.splitlines():
            if line.startswith('{"frisk_timing"'):
                split = json.loads(line)
        print(json.dumps({"run": os.path.basename(fa), "label": label, "rc": out.returncode, "wall_s": round(wall, 2),
                          "in_process_s": split and round(split["total_s"], 3),
                          "features": {f: gff_lines(os.path.join(tmp, f)) for f in ("hmm.gff3", "anomalies.gff3", "RIP_annotation.gff3")},
                          "split": split, "err": out.stderr[-400:] if out.returncode else None}), flush=True)

for path in [fa] + ([gz] if os.path.exists(gz) else []):
    # second run: the packed sequence cache (<fasta>.frisk2bit) and the genome pickle of the first are there; the windows are
    # scored again (--recalcWin): no parse / inflate, no pack, 0.5 B per base over PCIe
    for label, extra in (("cold caches", ["--recalc", "--recalcWin"]), ("sequence cache + genome pickle", ["--recalcWin"])):
        tmp = os.path.join(work, "T_" + os.path.basename(path))
        cmd = [sys.executable, "-m", "frisk_amd", "-H", path, "-k", "8", "-w", "5000", "-i", "1000", "-t", tmp,
               "--exitAfter", "WindowKLD"] + extra
        env = dict(os.environ, FRISK_TIMING="1", PYTHONPATH=ROOT)
        t0 = time.time()
        out = subprocess.run(cmd, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, cwd=ROOT)
        wall = time.time() - t0
        split = None
        for line in out.stderr.splitlines():
            if line.startswith('{"frisk_timing"'):
                split = json.loads(line)
        rows = sum(1 for _ in open(os.path.join(tmp, "raw_window_scores.bed"))) - 1 if out.returncode == 0 else -1
        import hashlib
        digest = hashlib.sha256(open(os.path.join(tmp, "raw_window_scores.bed"), "rb").read()).hexdigest()[:16] if out.returncode == 0 else None
        print(json.dumps({"run": os.path.basename(path), "label": label, "rc": out.returncode, "wall_s": round(wall, 2), "rows": rows, "table_sha256_16": digest,
                          "split": split, "err": out.stderr[-300:] if out.returncode else None}), flush=True)

# E2E_RANKS=N: the same job as N real ranks under torchrun on THIS one GPU (FRISK_DIST_REHEARSAL=1: gloo in RCCL's place) - a
# functional run of the sharded path at full size (tiles per rank, all-reduce, 3 M rows gathered on rank 0); its table must carry
# the digest of the one-process table above.  First run parses (and leaves the seek index), the second reads tiles through it.
if os.environ.get("E2E_RANKS"):
    import hashlib
    nr = int(os.environ["E2E_RANKS"])
    tmp = os.path.join(work, "R%d_" % nr + os.path.basename(fa))
    for label, extra in (("%d ranks on one GPU (rehearsal), cold caches" % nr, ["--recalc", "--recalcWin"]),
                         ("%d ranks on one GPU (rehearsal), seek index, profile recomputed" % nr, ["--recalcWin"])):
        if "seek index" in label:
            for f in os.listdir(tmp):
                if f.endswith("_genome.p"):
                    os.remove(os.path.join(tmp, f))          # (so that phase A runs again, through the index this time)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nr), "--master-addr", "127.0.0.1",
               "--master-port", "29731", "-m", "frisk_amd", "-H", fa, "-k", "8", "-w", "5000", "-i", "1000", "-t", tmp,
               "--exitAfter", "WindowKLD"] + extra
        env = dict(os.environ, FRISK_TIMING="1", FRISK_DIST_REHEARSAL="1", PYTHONPATH=ROOT)
        t0 = time.time()
        out = subprocess.run(cmd, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, cwd=ROOT)
        wall = time.time() - t0
        split = None
        for line in out.stderr.splitlines():
            if line.startswith('{"frisk_timing"'):
                split = json.loads(line)
        tab = os.path.join(tmp, "raw_window_scores.bed")
        ok = out.returncode == 0 and os.path.exists(tab)
        print(json.dumps({"run": os.path.basename(fa), "label": label, "rc": out.returncode, "wall_s": round(wall, 2),
                          "rows": sum(1 for _ in open(tab)) - 1 if ok else -1,
                          "table_sha256_16": hashlib.sha256(open(tab, "rb").read()).hexdigest()[:16] if ok else None,
                          "split_rank0": split, "index_used": "through the index" in out.stderr,
                          "err": out.stderr[-600:] if out.returncode else None}), flush=True)

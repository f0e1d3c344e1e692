"""`frisk` command line on the GPU hot path (SURVEY.md section 8, rows a11/a12 + f2 + f1).

Same option surface as the reference's argparse block (frisk/__init__.py L1127-1398: names, types, defaults,
choices, the inverted store_false semantics of --recalc/--recalcWin), same files in --tempDir (score table,
caches, GFF3) and the same row echo on stdout.  Phase A and phase B run through libfrisk_hip.so
(frisk_amd.hotpath); thresholds, merging and GFF3 writing are host numpy (frisk_amd.postprocess).

--hmmKLD runs frisk_amd.hmm (own 2-state Gaussian HMM; hmmlearn is absent and seeds randomly - parity unpinned).
Out of scope here: --runProjection/--cluster (sklearn analysis on a few hundred rows),
--graphics (seaborn/matplotlib), --gffIn intersections (bedtools).
Those options are accepted, as in the reference, and reported as unavailable if used.

Run under `python -m torch.distributed.run --nproc-per-node N -m frisk_amd ...` to shard one job over N GPUs
(frisk_amd.distributed): rank 0 writes the outputs.
"""
import argparse
import logging
import os
import pickle
import sys

import numpy as np

from . import postprocess as pp

log = logging.getLogger("frisk")


def build_parser():
    p = argparse.ArgumentParser(prog="frisk", description="Calculate all kmers in a given sequence")
    p.add_argument("--version", action="version", version="frisk --" + pp.FRISK_VERSION)
    # inputs
    p.add_argument("-H", "--hostSeq", type=str, required=True, help="host genome FASTA (one species)")
    p.add_argument("-Q", "--querySeq", type=str, default=None, help="scan this FASTA against the host profile (default: the host)")
    p.add_argument("--gffIn", type=str, default=None, help="GFF annotation of the scanned genome")
    # outputs
    p.add_argument("-O", "--outfile", type=str, default="raw_window_scores.bed", help="per-window score table, written in --tempDir")
    p.add_argument("-t", "--tempDir", type=str, default="temp", help="working / output directory")
    p.add_argument("--gffOutfile", type=str, default=None, help="GFF3 of merged anomalous features")
    p.add_argument("--hmmOutfile", type=str, default="2StateHmm.gff3", help="GFF3 of HMM state features")
    p.add_argument("--graphics", type=str, default=None, help="PDF of summary graphics")
    # output options
    p.add_argument("--mergeDist", type=int, default=0, help="merge anomalies within this many bases")
    p.add_argument("--gffFeatures", type=str, default=None, nargs="+", help="feature types of --gffIn to intersect with anomalies")
    p.add_argument("--gffRange", type=int, default=0, help="report annotations within this distance of anomalies")
    # core settings
    p.add_argument("-m", "--minWordSize", type=int, default="1", help="shortest k-mer")
    p.add_argument("-k", "--maxWordSize", type=int, default="8", help="longest k-mer")
    p.add_argument("-w", "--windowlen", type=int, default="5000", help="window length")
    p.add_argument("-i", "--increment", type=int, default="2500", help="window step")
    # run settings
    p.add_argument("--maskHost", action="store_true", default=False, help="skip soft-masked k-mers when profiling the host")
    p.add_argument("--exitAfter", default=None, choices=[None, "GenomeKmers", "WindowKLD"], help="stop after this stage")
    p.add_argument("--recalc", action="store_false", default=True, help="force recomputation of the host k-mer profile")
    p.add_argument("--recalcWin", action="store_false", default=True, help="force recomputation of the window scores")
    p.add_argument("--scaffoldsAll", action="store_true", default=False, help="score scaffolds below the minimum size as one window")
    # KLD thresholds
    p.add_argument("--threshTypeKLD", default=None, choices=[None, "percentile", "otsu"], help="how to pick the log10(KLD) cut")
    p.add_argument("--percentileKLD", type=float, default=99.0, help="percentile for --threshTypeKLD percentile")
    p.add_argument("--hmmKLD", action="store_true", default=False, help="2-state HMM segmentation of the KLD track")
    p.add_argument("-F", "--forceThresholdKLD", type=float, default=None, help="raw KLD above which a window is anomalous")
    # RIP
    p.add_argument("--RIP", action="store_true", default=False, help="report RIP indices per window and RIP features")
    p.add_argument("--RIPgff", type=str, default="RIP_annotation.gff3", help="GFF3 of RIP features")
    p.add_argument("--minCRI", type=float, default=0.0)
    p.add_argument("--peakCRI", type=float, default=1.0)
    p.add_argument("--minPI", type=float, default=1.0)
    p.add_argument("--maxSI", type=float, default=1.0)
    # projection / clustering (accepted, not available in this build)
    p.add_argument("--runProjection", default=None, choices=[None, "PCA", "PY-TSNE", "SKL-TSNE", "IncrementalPCA", "NMF", "MDS"])
    p.add_argument("--projectionDims", type=int, default=2)
    p.add_argument("--dimReduce", default="windows", choices=["features", "windows"])
    p.add_argument("--cluster", default=None, choices=[None, "DBSCAN", "KMEANS", "SPECTRAL"])
    p.add_argument("--dumpPCAdata", action="store_true", default=False)
    p.add_argument("--spikeNormal", action="store_true", default=False)
    p.add_argument("--pcaMin", type=int, default="1")
    p.add_argument("--pcaMax", type=int, default="6")
    p.add_argument("--perplexity", type=float, default=20.0)
    p.add_argument("--tsneGradient", default="barnes_hut", choices=["barnes_hut", "exact"])
    p.add_argument("--tsneInitPCA", default="random", choices=["random", "pca"])
    p.add_argument("--epsDBSCAN", type=float, default=10)
    p.add_argument("--kClusters", type=int, default=2)
    p.add_argument("--seed", default=None)
    p.add_argument("--chrmlist", default=None, nargs="+")
    p.add_argument("--updateHMM", action="store_true", default=False)
    p.add_argument("--updateWin", type=int, default=1000)
    p.add_argument("--updateInc", type=int, default=500)
    p.add_argument("--findSelf", action="store_true", default=False, help="report windows BELOW the threshold instead")
    return p


def mainArgs(argv=None):
    args = build_parser().parse_args(argv)
    if args.minWordSize > args.maxWordSize:
        logging.error("[ERROR] Minimum kmer size (-m/--minWordSize) must be less than Maximum kmer size (-k/--maxWordSize)\n")
        sys.exit(1)
    return args


def makePicklePath(args, space):
    """Cache file names of the reference (L497-506)."""
    base = os.path.basename(args.hostSeq)
    if space == "genome":
        return os.path.join(args.tempDir, "%s_kmers_%s_%s_genome.p" % (base, args.minWordSize, args.maxWordSize))
    if args.querySeq:
        base = os.path.basename(args.querySeq)
    return os.path.join(args.tempDir, "%s_kmers_%s_%s_KLD_window_%s_increment_%s.p"
                        % (base, args.minWordSize, args.maxWordSize, args.windowlen, args.increment))


def _columns(args):
    if args.RIP and args.minWordSize <= 2:
        return ["name", "start", "stop", "windowKLD", "GC", "PI", "SI", "CRI"]
    return ["name", "start", "stop", "windowKLD", "GC"]


def _fmt():
    return pp.py3_str if os.environ.get("FRISK_FLOAT_REPR", "py2") == "py3" else pp.py2_str


def write_table(path, columns, rows, echo=True):
    """Score table exactly as L1475-1494 lays it out: header, one tab-separated row per window, each row also
    printed.  Floats as Python 2's str() prints them (12 significant digits) unless FRISK_FLOAT_REPR=py3."""
    fmt = _fmt()
    with open(path, "w") as fh:
        fh.write("\t".join(columns) + "\n")
        for r in rows:
            line = "\t".join(fmt(v) for v in r)
            fh.write(line + "\n")
            if echo:
                print(line)


def main(argv=None):
    """Entry point.  A process group that this function creates is torn down before it returns."""
    import torch.distributed as dist
    had_group = dist.is_available() and dist.is_initialized()
    try:
        return _main(argv)
    finally:
        if not had_group and dist.is_available() and dist.is_initialized():
            dist.destroy_process_group()


def _main(argv=None):
    logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(funcName)s - %(message)s")
    args = mainArgs(argv)
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # the sharded code path can be forced in a one-rank job (tests rehearse it on a single GPU)
    sharded = world > 1 or os.environ.get("FRISK_FORCE_SHARDED") == "1"
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if sharded and not dist.is_initialized():
        import torch
        torch.cuda.set_device(local_rank)
        if world == 1:      # FRISK_FORCE_SHARDED outside a launcher: a one-rank rendezvous of our own
            for key, val in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29517"), ("RANK", "0"), ("WORLD_SIZE", "1")):
                os.environ.setdefault(key, val)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if rank == 0:
        print("frisk --", pp.FRISK_VERSION)
    genomepickle = makePicklePath(args, "genome")
    windowsPickle = makePicklePath(args, "window")
    querySeq = args.querySeq or args.hostSeq
    if rank == 0 and not os.path.isdir(os.path.abspath(args.tempDir)):
        os.makedirs(os.path.abspath(args.tempDir))
    for opt, why in (("cluster", "sklearn clustering is out of scope"),
                     ("graphics", "plotting is out of scope"), ("gffIn", "bedtools intersections are out of scope")):
        if getattr(args, opt):
            log.warning("--%s is not available in this build: %s", opt, why)

    from .hotpath import HotPath, mapsToProfile
    columns = _columns(args)
    rip = len(columns) == 8
    hp = HotPath(args.minWordSize, args.maxWordSize, device=local_rank)
    try:
        # ---- phase A: host k-mer profile (L1436-1447); --recalc is store_false: giving it forces recomputation
        if os.path.isfile(genomepickle) and args.recalc and not sharded:
            log.info("Importing previously calculated genome kmers from %s", genomepickle)
            with open(genomepickle, "rb") as fh:
                genomeKmers = pickle.load(fh, encoding="latin1")
            hp.setGenomeProfile(genomeKmers)
        else:
            log.info("Calculating kmers for host sequence: %s", args.hostSeq)
            if sharded:
                rows_all = _sharded(hp, args, querySeq, rip)
            else:
                genomeKmers = hp.genomeProfile(args)
                with open(genomepickle, "wb") as fh:
                    pickle.dump(genomeKmers, fh, protocol=2)
            if args.exitAfter == "GenomeKmers":
                log.info("Finished counting kmers. Exiting.")
                return 0
        # ---- phase B: window scores (L1454-1507)
        if not sharded and os.path.isfile(windowsPickle) and args.recalcWin:
            log.info("Importing previously calculated window KLD scores from: %s", windowsPickle)
            with open(windowsPickle, "rb") as fh:
                cached = pickle.load(fh)
            if hasattr(cached, "columns") and hasattr(cached, "itertuples"):       # a DataFrame, as the reference pickles
                if "windowKLD" not in cached.columns:                               # legacy column name (L1458-1459)
                    cached["windowKLD"] = cached["windowKLI"]
                rows = [tuple(r) for r in cached[columns].itertuples(index=False, name=None)]
            else:
                rows = [tuple(r) for r in cached["rows"]]
        else:
            if sharded:
                rows = rows_all
            else:
                rows, _ = hp.scanGenome(args, querySeq)
            if rank == 0:
                write_table(os.path.join(args.tempDir, args.outfile), columns, rows)
                with open(windowsPickle, "wb") as fh:
                    pickle.dump({"columns": columns, "rows": rows}, fh, protocol=2)
    finally:
        hp.close()
    if rank != 0:
        return 0
    if args.exitAfter == "WindowKLD":
        log.info("Finished calculating window KLD scores. Exiting.")
        return 0

    # ---- thresholds and features (L1522-1530, L1671-1707)
    allKLD = np.array([[float(r[3])] for r in rows], dtype=float)
    with np.errstate(divide="ignore"):
        logKLD = np.log10(allKLD)
    threshold, _bins = pp.setKLDThresh(args, logKLD)
    threshold = float(np.ravel(threshold)[0])
    log.info("log10(KLD) threshold = %s", threshold)
    if args.hmmKLD:                                                     # L1537-1548
        from .hmm import hmm2BED, hmmBED2GFF
        intervals, _model = hmm2BED(rows)
        with open(os.path.join(args.tempDir, args.hmmOutfile), "w") as fh:
            for line in hmmBED2GFF(intervals):
                fh.write(line)
    if args.runProjection:                                              # L1556-1596: counts for the projection
        from .fasta import readFasta
        from .projection import symmetricCounts
        feats, _ = pp.thresholdKLD(rows, threshold, args, merge=(args.dimReduce == "features"))
        names, seqs = readFasta(querySeq)
        fasta = dict(zip(names, seqs))
        labelled = [(":".join([f[0], str(f[1]), str(f[2])]), fasta[f[0]][int(f[1]) - 1:int(f[2])]) for f in feats if f[0] in fasta]
        anomLabels, anomCounts = symmetricCounts(labelled, args.pcaMin, args.pcaMax, device=local_rank)
        if args.dumpPCAdata:
            with open(os.path.join(args.tempDir, "anomLabels"), "wb") as fh:
                pickle.dump(anomLabels, fh, protocol=2)
            with open(os.path.join(args.tempDir, "anomCounts"), "wb") as fh:
                pickle.dump(anomCounts, fh, protocol=2)
        log.info("Symmetric k-mer proportions of %s anomalous windows computed; the %s projection is not built here.",
                 len(labelled), args.runProjection)
    anomalies, _sel = pp.thresholdKLD(rows, threshold, args, merge=True)
    log.info("Detected %s features above KLD threshold.", len(anomalies))
    if args.gffOutfile:
        with open(os.path.join(args.tempDir, args.gffOutfile), "w") as fh:
            for line in pp.anomaly2GFF(anomalies, args):
                fh.write(line)
    if rip:
        feats = pp.thresholdRIP(rows, args)
        if feats:
            with open(os.path.join(args.tempDir, args.RIPgff), "w") as fh:
                for line in pp.RIP2GFF(feats):
                    fh.write(line)
        else:
            log.info("No RIP features detected.")
    return 0


def _sharded(hp, args, querySeq, rip):
    """N ranks, one job: frisk_amd.distributed.run_sharded_files (every rank reads the FASTA natively and takes an
    equal share of the positions and of the candidate windows); returns rows (rank 0) in table form."""
    from .distributed import run_sharded_files
    rows = run_sharded_files(hp.engine, args.hostSeq, args.windowlen, args.increment, mask_host=args.maskHost, rip=rip,
                             scaffolds_all=args.scaffoldsAll, query_path=querySeq)
    if rows is None:
        return None
    from . import _ffi
    out = []
    for r in rows:
        if r[3] & _ffi.ROW_ZERO_WEIGHT:
            raise ZeroDivisionError("float division by zero")
        kld = 0 if (r[3] & _ffi.ROW_NO_MAXMER) else r[4]
        out.append((r[0], r[1], r[2], kld, r[5]) + tuple(r[6:]))
    return out

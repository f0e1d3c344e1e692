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


def write_table(path, table, echo=True):
    """Score table exactly as L1475-1494 lays it out: header, one tab-separated row per window, each row also
    printed.  Floats as Python 2's str() prints them (12 significant digits) unless FRISK_FLOAT_REPR=py3.
    The text comes from the library's native formatter in one piece (ScoreTable.text): no per-row Python."""
    fmt = None if _fmt() is pp.py2_str else _fmt()
    raw = getattr(sys.stdout, "buffer", None) if echo else None
    if fmt is None and (raw is not None or not echo):
        # the formatter's bytes go to the file and to stdout as they are (no decode / encode of a few hundred megabytes)
        body = table.text_bytes()
        with open(path, "wb") as fh:
            fh.write(("\t".join(table.columns) + "\n").encode("utf-8"))
            fh.write(body)
        if echo:
            sys.stdout.flush()
            raw.write(body)
            raw.flush()
        return
    body = table.text(fmt)
    with open(path, "w") as fh:
        fh.write("\t".join(table.columns) + "\n")
        fh.write(body)
    if echo:
        sys.stdout.write(body)


def _load_window_cache(path, rip):
    """The window cache the reference pickles is the allWindows DataFrame (L1501); older builds of this package wrote
    {columns, rows}.  Both are read."""
    from .table import ScoreTable
    with open(path, "rb") as fh:
        cached = pickle.load(fh)
    if hasattr(cached, "columns") and hasattr(cached, "itertuples"):
        return ScoreTable.from_frame(cached, rip)
    return ScoreTable.from_rows([tuple(r) for r in cached["rows"]], rip=rip)


def _dump_window_cache(path, table):
    """As the reference: pickle.dump(allWindows DataFrame) (L1501) - so that any tool that reads *_KLD_window_*.p with a current
    pandas can load it.  (Protocol 2 was used here until round 3: under Python 3 it sends
    every numeric column through a latin-1 text detour - 0.5 s per 3 M rows against 0.15 - and no Python 2 pandas reads a frame
    pickled by today's pandas anyway.)  Without pandas: the {columns, rows} form."""
    try:
        obj = table.to_frame()
    except ImportError:
        obj = {"columns": table.columns, "rows": table.rows()}
    with open(path, "wb") as fh:
        # (protocol 5 hands the numeric columns over as buffers instead of copying them into the stream: 0.16 s against 0.24 per
        #  3 M rows, held under the interpreter lock either way; any Python >= 3.8 reads it)
        pickle.dump(obj, fh, protocol=pickle.HIGHEST_PROTOCOL)


def main(argv=None):
    """Entry point.  A process group that this function creates is torn down before it returns."""
    import torch.distributed as dist
    had_group = dist.is_available() and dist.is_initialized()
    try:
        return _main(argv)
    finally:
        if not had_group and dist.is_available() and dist.is_initialized():
            dist.destroy_process_group()


class _Clock:
    """Wall-clock split of one run (logged at the end; FRISK_TIMING=1 also prints it as one JSON line on stderr)."""

    def __init__(self):
        import time
        self._now = time.perf_counter
        self.t0 = self._now()
        self.last = self.t0
        self.parts = []

    def lap(self, label):
        now = self._now()
        self.parts.append((label, now - self.last))
        self.last = now

    def report(self):
        total = self._now() - self.t0
        log.info("timing: " + ", ".join("%s %.3f s" % p for p in self.parts) + ", total %.3f s" % total)
        if os.environ.get("FRISK_TIMING") == "1":
            import json
            sys.stderr.write(json.dumps({"frisk_timing": dict(self.parts), "total_s": total}) + "\n")


def _main(argv=None):
    logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(funcName)s - %(message)s")
    args = mainArgs(argv)
    clock = _Clock()
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # the sharded code path can be forced in a one-rank job (tests rehearse it on a single GPU)
    sharded = world > 1 or os.environ.get("FRISK_FORCE_SHARDED") == "1"
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # FRISK_DIST_REHEARSAL=1: the N-rank job on ONE GPU - every rank on device 0, gloo instead of RCCL (which refuses two ranks
    # on one device): a functional rehearsal of the sharded path (tiles, all-reduce, row gather) where only one GPU exists
    rehearsal = os.environ.get("FRISK_DIST_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    if sharded and not dist.is_initialized():
        import torch
        torch.cuda.set_device(local_rank)
        if world == 1:      # FRISK_FORCE_SHARDED outside a launcher: a one-rank rendezvous of our own
            for key, val in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29517"), ("RANK", "0"), ("WORLD_SIZE", "1")):
                os.environ.setdefault(key, val)
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if rank == 0:
        print("frisk --", pp.FRISK_VERSION)
    genomepickle = makePicklePath(args, "genome")
    windowsPickle = makePicklePath(args, "window")
    querySeq = args.querySeq or args.hostSeq
    if rank == 0 and not os.path.isdir(os.path.abspath(args.tempDir)):
        os.makedirs(os.path.abspath(args.tempDir))
    if sharded:
        dist.barrier()      # tempDir exists, and every rank sees the same cache files, before anyone looks for them
    for opt, why in (("cluster", "sklearn clustering is out of scope"),
                     ("graphics", "plotting is out of scope"), ("gffIn", "bedtools intersections are out of scope")):
        if getattr(args, opt):
            log.warning("--%s is not available in this build: %s", opt, why)

    from .hotpath import HotPath
    from . import distributed as D
    columns = _columns(args)
    rip = len(columns) == 8
    w, inc, all_ = args.windowlen, args.increment, bool(args.scaffoldsAll)
    # (the packed-sequence cache lives beside the pickle caches; --recalc - store_false - means: recompute, so rewrite it too)
    hp = HotPath(args.minWordSize, args.maxWordSize, device=local_rank, cache_dir=None if sharded else args.tempDir,
                 use_cache=bool(args.recalc))
    clock.lap("startup (imports + HIP context)")
    table = None
    index_writers = []
    out_threads, out_err = [], []           # the score table text and the window cache on their way to disk

    def _join_outputs():
        for th in out_threads:
            th.join()
        del out_threads[:]
        if out_err:
            raise out_err.pop(0)

    def _bg(fn, *a):
        import threading

        def run():
            try:
                fn(*a)
            except BaseException as err:        # noqa: B902 - re-raised on the main thread (_join_outputs)
                out_err.append(err)
        th = threading.Thread(target=run, name="frisk-" + fn.__name__)
        th.start()
        out_threads.append(th)

    def _write_genome_pickle(arrays):
        # the reference's pickled form (list of dicts + 3 metadata dicts, L356-359): 87 380 dict entries at K = 8, built and written
        # beside the scan, which needs only the profile on the device
        from .hotpath import profileToMaps
        maps = profileToMaps(arrays[0], arrays[1], arrays[2], arrays[3], args.minWordSize, args.maxWordSize)
        tmp = genomepickle + ".tmp%d" % os.getpid()
        with open(tmp, "wb") as fh:
            pickle.dump(maps, fh, protocol=2)
        os.replace(tmp, genomepickle)

    # Sharded jobs: a rank with a seek index of the FASTA (fasta_index.h; in --tempDir, or beside the file) copies the bytes of
    # its tiles instead of parsing all of it.  --recalc (store_false: recompute) ignores every index; where a rank had to
    # parse, rank 0 writes the index beside the other caches for the next run - in the background, it is not needed by this one.
    def _shard_index(fasta):
        from .fasta import fastaIndexPaths
        paths = fastaIndexPaths(fasta, args.tempDir)
        return paths if args.recalc else []         # (--recalc: no index at all, not ours and not a foreign <fasta>.fai)

    def _note_shard_load(fasta):
        if hp.engine.shard_index is not None:
            log.info("Read this rank's tiles of %s through the index %s", fasta, hp.engine.shard_index)
        elif rank == 0:
            import threading
            from .fasta import fastaIndexPaths, writeFastaIndex

            def _write(src=fasta, dst=fastaIndexPaths(fasta, args.tempDir)[0]):
                try:
                    writeFastaIndex(src, dst)
                except Exception as err:            # (a read-only --tempDir, a vanished file: the next run parses again)
                    log.info("No seek index written for %s: %s", src, err)
            th = threading.Thread(target=_write, name="frisk-fasta-index", daemon=True)
            th.start()
            index_writers.append(th)

    try:
        # ---- phase A: host k-mer profile (L1436-1447); --recalc is store_false: giving it forces recomputation.
        # Under torchrun every rank takes the same branch (the cache is a file all ranks see); rank 0 writes it.
        resident_names = None       # sharded: names of the FASTA whose tiles are resident
        if os.path.isfile(genomepickle) and args.recalc:
            log.info("Importing previously calculated genome kmers from %s", genomepickle)
            with open(genomepickle, "rb") as fh:
                genomeKmers = pickle.load(fh, encoding="latin1")
            hp.setGenomeProfile(genomeKmers)
            clock.lap("profile cache")
        else:
            log.info("Calculating kmers for host sequence: %s", args.hostSeq)
            if sharded:
                resident_names = D.profile_sharded(hp.engine, args.hostSeq, w, inc, mask_host=args.maskHost, scaffolds_all=all_,
                                                   index=_shard_index(args.hostSeq))
                _note_shard_load(args.hostSeq)
                profile_arrays = hp.engine.profile_get() if rank == 0 else None
            else:
                hp._load(args.hostSeq)
                clock.lap("packed sequence cache -> HBM" if hp.loaded_from_cache else "FASTA parse + upload + pack")
                profile_arrays = hp.genomeProfileArrays(args)
            clock.lap("phase A (profile)" if not sharded else "phase A (parse + upload + profile + all-reduce)")
            if rank == 0:
                _bg(_write_genome_pickle, profile_arrays)               # (joined with the other outputs, or right below)
            if args.exitAfter == "GenomeKmers":                          # L1443-1445: before any window is scored
                _join_outputs()
                clock.lap("profile pickle")
                log.info("Finished counting kmers. Exiting.")
                return 0
        # ---- phase B: window scores (L1454-1507)
        if os.path.isfile(windowsPickle) and args.recalcWin:
            log.info("Importing previously calculated window KLD scores from: %s", windowsPickle)
            if rank == 0:
                table = _load_window_cache(windowsPickle, rip)
            clock.lap("window cache")
        else:
            zero = None
            try:
                if sharded:
                    same = querySeq == args.hostSeq and resident_names is not None
                    table, failed = D.scan_sharded(hp.engine, querySeq, w, inc, rip=rip, scaffolds_all=all_,
                                                   resident_names=resident_names if same else None, index=_shard_index(querySeq))
                    if not same:
                        _note_shard_load(querySeq)
                    if failed:
                        zero = ZeroDivisionError("float division by zero")          # on EVERY rank (L437)
                else:
                    table, _res = hp.scanTable(args, querySeq)
            except ZeroDivisionError as err:
                zero, table = err, getattr(err, "table", None)
            clock.lap("phase B (scan)")
            if rank == 0 and table is not None:
                if zero is None:
                    # Two files of the same rows - the window cache (L1501) and the table text (native formatter, outside the
                    # interpreter lock), neither reads the other - written in the background while thresholds, segmentation and
                    # features are computed from the same (read-only) columns; joined before this function returns or raises.
                    _bg(_dump_window_cache, windowsPickle, table)
                    _bg(write_table, os.path.join(args.tempDir, args.outfile), table)
                else:
                    # the reference writes and prints row by row (L1487-1494): what it had written before dying is written here too
                    write_table(os.path.join(args.tempDir, args.outfile), table)
                clock.lap("score table text + window pickle started" if zero is None else "score table text")
            if zero is not None:
                raise zero
    except BaseException:
        for th in out_threads:
            th.join()
        raise
    finally:
        for th in index_writers:
            th.join()
        hp.close()
    if rank != 0:
        return 0
    if args.exitAfter == "WindowKLD":
        _join_outputs()
        clock.lap("score table text + window pickle written")
        log.info("Finished calculating window KLD scores. Exiting.")
        clock.report()
        return 0

    try:
        # ---- thresholds and features (L1522-1530, L1671-1707)
        kld = np.where(table.kld_is_int0 != 0, 0.0, table.kld)
        with np.errstate(divide="ignore"):
            logKLD = np.log10(kld.reshape(-1, 1))
        threshold, _bins = pp.setKLDThresh(args, logKLD)
        threshold = float(np.ravel(threshold)[0])
        log.info("log10(KLD) threshold = %s", threshold)
        clock.lap("KLD threshold")
        if args.hmmKLD:                                                     # L1537-1548
            from .hmm import hmm2BED, hmmBED2GFF
            intervals, _model = hmm2BED(table)
            with open(os.path.join(args.tempDir, args.hmmOutfile), "w") as fh:
                fh.writelines(hmmBED2GFF(intervals))
            log.info("HMM segmentation: %s state features (fit: %s EM rounds, log-likelihood %s)", len(intervals),
                     getattr(_model, "n_iter_", "?"), getattr(_model, "loglik_", "?"))
            clock.lap("HMM segmentation + GFF")
        if args.runProjection:                                              # L1556-1596: counts for the projection
            from .fasta import readFasta
            from .projection import symmetricCounts
            feats, _ = pp.thresholdKLD(table, threshold, args, merge=(args.dimReduce == "features"))
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
        anomalies, _sel = pp.thresholdKLD(table, threshold, args, merge=True)
        log.info("Detected %s features above KLD threshold.", len(anomalies))
        if args.gffOutfile:
            with open(os.path.join(args.tempDir, args.gffOutfile), "w") as fh:
                for line in pp.anomaly2GFF(anomalies, args):
                    fh.write(line)
        if rip:
            feats = pp.thresholdRIP(table, args)
            if feats:
                with open(os.path.join(args.tempDir, args.RIPgff), "w") as fh:
                    for line in pp.RIP2GFF(feats):
                        fh.write(line)
            else:
                log.info("No RIP features detected.")
    except BaseException:
        for th in out_threads:           # (the two output files are finished before the error travels on)
            th.join()
        raise
    clock.lap("anomaly / RIP features + GFF")
    _join_outputs()
    clock.lap("score table text + window pickle written")
    clock.report()
    return 0

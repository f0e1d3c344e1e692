#!/usr/bin/env python3
"""bench.py - whole-job throughput of the frisk hot path on N MI355X (one process per GPU).

A step = one pass of the hot path over the rank's resident synthetic assembly:
    phase A  profile_reset -> profile_add -> [ONE all-reduce of the raw profile over RCCL] -> finalize
    phase B  window scan of every candidate window (kernels + D2H of the result rows)
with the packed scaffolds already resident in HBM when the timed region starts (`value`; the task's bench contract).

Workload (config.workload): BASELINE.json's metric geometry k=1..8, w=5000, i=1000 on the configuration the metric is quoted
on - C5, the GRCh38-like shape (24 chromosome-scale + 400 small synthetic scaffolds, 3.29 Gb, ~7 % N, 3.28 M candidate windows):
it fits one GPU (4.9 GB of 288), so N = 1 is the whole C5 job on one MI355X.  Weak scaling: every rank owns one assembly of
that shape (its own seed), the genome profile is pooled over all ranks by the one all-reduce.  `strong` (N > 1) is the other
reading of BASELINE's C5 line: ONE assembly split over the N ranks.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment launches its own N ranks
(torch.distributed.run, 127.0.0.1) before anything touches the GPU and relays rank 0's line.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     - HBM roofline of the dominant kernel (scan): algorithmic bytes / HIP-event kernel time; `traffic`
                 and the `binding` block (what actually binds: VALU issue, LDS, waits) come from rocprofv3 PMC passes
                 of the same workload collected OFFLINE and committed under profiles/ (named in the line)
  inclusive    - SURVEY.md 8(d)'s quantity: the SAME whole job from page-locked host memory, timed over H2D + profile + scan +
                 D2H, cold (a fresh batch every time): the 0.25 B/base form (2-bit codes + run lists of the two masks, the
                 codes streamed in pieces with phase A following them), beside the 0.5 B/base and the ASCII forms (N = 1 only)
  cold         - the FIRST step on a freshly resident batch (the adaptive counter width's sample is paid there)
  shard        - the 1/8 LPT shard of the C5 shape that rounds 1-3 reported as `value` (the per-GPU work of an 8-GPU C5 job)
  strong       - N > 1: ONE C5 assembly split over the N ranks
  realistic    - the shard with repeat content (simple repeats of period 1-6, satellite arrays), unmasked and soft-masked:
                 the counter widths real assemblies take, first (cold) step included
  cpu_baseline - the reference-shaped Python oracle timed on one host core on a bounded sample
                 (+ cpu_baseline_numpy, cpu_baseline_c: the vectorised and the compiled multi-thread restatements)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

KMIN, KMAX, W, INC = 1, 8, 5000, 1000
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def host_cpu():
    """(model name, logical CPUs of the box, CPUs this process may run on)."""
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count()
    return model, os.cpu_count(), usable


def cpu_baseline(engine, names_seq0_len, n_windows):
    """Time oracle/frisk_oracle.py (the reference-shaped Python restatement, 1 core) on the first
    `n_windows` candidate windows of scaffold 0, against the profile the GPU just built.
    Returns (dict for the JSON line, max |KLD_gpu - KLD_oracle| on the sample, the numpy oracle's line, the
    compiled C oracle's line)."""
    import numpy as np
    from oracle import frisk_oracle as O
    from frisk_amd.hotpath import profileToMaps

    sym, tl, ex, nn = engine.profile_get()
    gmaps = profileToMaps(sym, tl, ex, nn, KMIN, KMAX)[:KMAX - KMIN + 1]
    gmeta = {"totalLen": tl, "exMax": ex, "nnTotal": nn}
    n_windows = min(n_windows, max(0, (names_seq0_len - W) // INC + 1))
    span = W + (n_windows - 1) * INC
    seq = engine.read_seq(0, 0, span).decode("ascii")
    res = engine.scan(W, INC, c0=0, c1=n_windows)
    t0 = time.perf_counter()
    rows = []
    for j in range(n_windows):                      # the regular candidates j = 0..n_windows-1 of scaffold 0
        win = seq[j * INC:j * INC + W]
        if O.count_acgt(win)[1] >= 0.3 * len(win):  # the reference's N filter (L237-241)
            continue
        rows.append(O.score_window(win, gmaps, gmeta, KMIN, KMAX))
    dt = time.perf_counter() - t0
    # second, separately labelled line: the vectorised numpy restatement on the same windows
    from oracle import frisk_oracle_np as N
    ig = N.genome_ivom_table(np.asarray(sym), (tl, ex, nn), KMIN, KMAX)
    enc = N.Encoded(seq)
    t1 = time.perf_counter()
    n_np = 0
    for j in range(n_windows):
        win = enc.slice(j * INC, j * INC + W)
        if (win.n - int(win.upper.sum())) >= 0.3 * win.n:
            continue
        N.score_window(win, ig, KMIN, KMAX)
        n_np += 1
    dt_np = time.perf_counter() - t1
    numpy_line = {"value": n_np / dt_np if dt_np > 0 else 0.0, "unit": "windows/s", "cores": 1, "kind": "port",
                  "sample": "oracle/frisk_oracle_np.py (vectorised numpy restatement, NOT the reference's structure) on the "
                            "same %d windows; %.2f s" % (n_np, dt_np)}
    c_line = cpu_baseline_c(engine, np.asarray(sym), (tl, ex, nn), names_seq0_len)
    kept = np.nonzero(res.kept)[0][:len(rows)]
    worst = max((abs(float(res.kld[r]) - row["KLD"]) for r, row in zip(kept.tolist(), rows)), default=0.0)
    return ({"value": len(rows) / dt if dt > 0 else 0.0, "unit": "windows/s", "cores": 1, "kind": "port",
             "sample": "oracle/frisk_oracle.py (reference-shaped Python, single thread) on the first %d kept "
                       "windows of scaffold 0 of the same synthetic shard, k=%d..%d w=%d i=%d; %.1f s"
                       % (len(rows), KMIN, KMAX, W, INC, dt)}, worst, numpy_line, c_line)


def cpu_baseline_c(engine, sym, meta, seq0_len, n_windows=60000):
    """Third, separately labelled CPU line: the compiled C + OpenMP oracle (oracle/frisk_oracle_c.c) on all host
    threads, on the first `n_windows` candidates of scaffold 0, checked row by row against the GPU."""
    import numpy as np
    from oracle import frisk_oracle_c as OC
    n_windows = min(n_windows, max(0, (seq0_len - W) // INC + 1))
    if n_windows <= 0:
        return None
    span = W + (n_windows - 1) * INC
    S = OC.Seqs([engine.read_seq(0, 0, span)])
    ig = OC.genome_ivom(sym, meta, KMIN, KMAX)
    OC.scan(S, ig, KMIN, KMAX, W, INC, cand=(0, 256))                 # page in, spin the thread pool up
    t0 = time.perf_counter()
    exp = OC.scan(S, ig, KMIN, KMAX, W, INC, cand=(0, n_windows))
    dt = time.perf_counter() - t0
    res = engine.scan(W, INC, c0=0, c1=n_windows)
    k = np.nonzero(res.kept)[0]
    same_rows = len(k) == len(exp["kld"]) and bool(np.array_equal(res.start[k], exp["start"]))
    worst = float(np.max(np.abs(res.kld[k] - exp["kld"]))) if same_rows and len(k) else float("nan")
    return {"value": len(exp["kld"]) / dt if dt > 0 else 0.0, "unit": "windows/s", "cores": int(OC.lib().fo_threads()),
            "kind": "port", "max_abs_dKLD_vs_gpu": worst, "rows_match_gpu": same_rows,
            "sample": "oracle/frisk_oracle_c.c (compiled C + OpenMP restatement, all host threads) on the first %d "
                      "candidate windows of scaffold 0 (%d kept); %.2f s" % (n_windows, len(exp["kld"]), dt)}


def inclusive_block(eng, lens, step, fence, steps, rows, bases):
    """windows/s = emitted rows / wall time of phase B's job from host memory - kernels + H2D + D2H, FASTA
    parsing (and the host-side packing, which the parser's threads do) excluded.  The genome profile needs EVERY base before
    the first window can be scored, so inside one job the scan cannot start before the upload ends; what CAN overlap is
    phase A: the codes cross PCIe in pieces and profile_add follows the pieces.  Every timed job is cold: a fresh batch in the
    other slot, the adaptive counter width sampled again.  Host buffers page-locked.  Three forms of the same job:
      from_2bit    0.25 B/base codes + run lists of the two masks (frisk_pack_2bit -> frisk_seq_stage_2bit)   <- the figure
      stream_of_jobs_2bit  the same for job after job: the NEXT job's upload runs under this job's scan (two batch slots), every
                   job still cold - what a queue of assemblies, or of query chunks against one host profile, gets
      from_packed  0.5 B/base: codes + two dense bitmaps (frisk_seq_stage_packed)
      from_ascii   1 B/base, packed on the device (frisk_seq_stage)"""
    import numpy as np
    total = sum(lens)
    big = eng.host_array("ascii", total)
    views, o = [], 0
    piece = 1 << 26
    for i, n in enumerate(lens):            # the synthetic assembly back to the host once (not timed)
        for a in range(0, n, piece):
            m = min(piece, n - a)
            big[o + a:o + a + m] = np.frombuffer(eng.read_seq(i, a, m), dtype=np.uint8)
        views.append(big[o:o + n])
        o += n
    t0 = time.perf_counter()
    codes2, inv_runs, low_runs, _ = eng.pack_2bit(views, pinned=True)
    pack_s = time.perf_counter() - t0
    codes, inv, low = eng.export_packed(pinned=True)
    out = {}

    def timed(body, label, nbytes, reps):
        body()                               # warm-up: allocates the second batch slot
        fence()
        t0 = time.perf_counter()
        for _ in range(reps):
            body()
        fence()
        dt = (time.perf_counter() - t0) / reps
        out[label] = {"windows_per_s": rows / dt, "gbases_per_s": bases / dt / 1e9, "ms_per_job": dt * 1e3,
                      "pcie_bytes_per_job": int(nbytes), "jobs_timed": reps}

    def job_2bit():
        eng.stage_2bit(codes2, inv_runs, low_runs, lens); eng.commit(); step()

    def job_packed():
        eng.stage_packed(codes, inv, low, lens); eng.commit(); step()

    def job_ascii():
        eng.stage(views); eng.commit(); step()

    def stream_2bit():                   # a STREAM of jobs: the next assembly's upload under this one's scan (commit = device-side wait)
        eng.stage_2bit(codes2, inv_runs, low_runs, lens); step(); eng.commit()

    timed(job_2bit, "from_2bit", codes2.nbytes + inv_runs.nbytes + low_runs.nbytes, steps)
    timed(stream_2bit, "stream_of_jobs_2bit", codes2.nbytes + inv_runs.nbytes + low_runs.nbytes, steps)
    timed(job_packed, "from_packed", codes.nbytes + inv.nbytes + low.nbytes, max(2, steps // 2))
    timed(job_ascii, "from_ascii", total, max(2, steps // 2))
    out["value"] = out["from_2bit"]["windows_per_s"]
    out["unit"] = "windows/s (emitted rows / wall time of H2D + profile + scan + D2H; every job cold)"
    out["mask_runs"] = {"inv": int(inv_runs.shape[0]), "low": int(low_runs.shape[0])}
    out["host_pack_s"] = pack_s
    out["note"] = ("SURVEY.md 8(d)'s definition of windows/s; never `value` (the bench contract keeps `value` on inputs resident in HBM).  "
                   "host_pack_s = frisk_pack_2bit over the whole assembly (ASCII -> 2 bits + run lists, host threads), part of parsing, not timed")
    return out

def csrc_hash():
    """sha256 over the library's sources (frisk_amd/csrc/*, include/*.h): a profile of the scan kernel is only quoted for the
    sources it was taken from (tools/pmc_bench_json.py stamps the same hash into profiles/*_pmc_bench.json)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(ROOT, "frisk_amd", "csrc", "*")) + glob.glob(os.path.join(ROOT, "include", "*.h"))):
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def timed_steps(step, fence, warmup, steps):
    """(seconds per step, last result, kernel times) - barrier + synchronize on both sides, as the headline loop."""
    for _ in range(warmup):
        step()
    fence()
    t0 = time.perf_counter()
    scan_ms = []
    for _ in range(steps):
        res, _tp, ts = step()
        scan_ms.append(ts)
    fence()
    return (time.perf_counter() - t0) / steps, res, sum(scan_ms) / len(scan_ms)


def cold_step(eng, synth_args, step, fence):
    """One step on a batch that has just become resident (re-generated: any per-batch state of the library is gone): the
    adaptive counter width runs its sample, with a host synchronisation, inside this step."""
    lens, kw = synth_args
    eng.synth(lens, **kw)
    fence()
    t0 = time.perf_counter()
    res, _tp, ts = step()
    fence()
    return (time.perf_counter() - t0) * 1e3, ts, res


def shape_block(eng, lens, kw, step, fence, steps, label):
    """One more shape on the same shard: cold first step, then warm steps."""
    import numpy as np
    cold_ms, cold_scan_ms, _ = cold_step(eng, (lens, kw), step, fence)
    dt, res, scan_ms = timed_steps(step, fence, 1, steps)
    rows = int(res.kept.sum())
    width, h8, h16, _seg = eng.scan_stat()
    n = len(res)
    return {"shape": label, "synth": {k: v for k, v in kw.items() if k != "seed"},
            "value": rows / dt, "unit": "windows/s (emitted rows, as `value`)", "candidate_windows_per_s": n / dt,
            "ms_per_step": dt * 1e3, "scan_kernel_ms": scan_ms, "rows": rows, "candidate_windows": n,
            "share_of_candidates_kept": rows / max(n, 1),
            "scan_counter_width": {"bulk_bits": width, "side_table_for_period4_maxmers": eng.scan_side(), "windows_handed_to_8bit": h8,
                                   "windows_handed_to_16bit": h16},
            "cold_first_step_ms": cold_ms, "cold_first_scan_kernel_ms": cold_scan_ms,
            "max_kld": float(np.nanmax(res.kld[res.kept])) if rows else None}


def c5_lens():
    """The whole C5 shape: the eight LPT shards one after the other (424 scaffolds, 3.29 Gb)."""
    from frisk_amd import synth
    return [n for r in range(8) for n in synth.c5_shard_lens(8, r)]


def shard_block(eng, rank, make_step, fence, steps):
    """The 1/8 LPT shard of the C5 shape (~410 Mb, ~410 k candidate windows) that rounds 1-3 reported as `value`: the per-GPU
    work of an 8-GPU C5 job, and the size at which a scan's fixed costs show."""
    from frisk_amd import synth
    lens = synth.c5_shard_lens(8, rank % 8)
    kw = dict(seed=0xC5 + rank, island_frac=0.02, n_frac=0.07, lower_frac=0.0)
    step = make_step()
    cold_ms, cold_scan_ms, _ = cold_step(eng, (lens, kw), step, fence)
    dt, res, scan_ms = timed_steps(step, fence, 2, steps)
    rows = int(res.kept.sum())
    return {"workload": "one 1/8 LPT shard of the C5 shape, resident", "bases": sum(lens), "candidate_windows": len(res), "rows": rows,
            "value": rows / dt, "unit": "windows/s", "ms_per_step": dt * 1e3, "scan_kernel_ms": scan_ms,
            "scan_kernel_windows_per_s": len(res) / (scan_ms * 1e-3), "cold_first_step_ms": cold_ms,
            "cold_first_scan_kernel_ms": cold_scan_ms}, lens


def strong_block(eng, dist, torch, rank, world, red_dev, fence, steps):
    """The whole C5 shape as ONE job over the N ranks: every rank holds the packed assembly (4.9 GB of 288), counts the k-mers
    that start in its N-th of the positions, the raw profiles are summed by the one all-reduce, and every rank scores its N-th
    of the candidate windows (contiguous ranges in output order; rows stay on the rank, as in the weak steps).  N = 1 is the
    whole job on one GPU.  (The CLI's multi-GPU path shards the residency too - frisk_fasta_load_shard; here the point is the
    time of the job against N.)"""
    lens = c5_lens()
    eng.synth(lens, seed=0xC5, island_frac=0.02, n_frac=0.07, lower_frac=0.0)
    n_cand = eng.scan_plan(W, INC)
    padded = eng.padded_len
    c0, c1 = n_cand * rank // world, n_cand * (rank + 1) // world
    p0, p1 = (padded // 32 * rank // world) * 32, (padded // 32 * (rank + 1) // world) * 32      # (whole 32-position words)
    if rank == world - 1:
        p1 = padded

    def step():
        eng.profile_reset()
        eng.profile_add(mask_host=False, pos_begin=p0, pos_end=p1)
        eng.profile_allreduce()
        eng.profile_finalize()
        res = eng.scan(W, INC, c0=c0, c1=c1, pinned=True)
        return res, eng.kernel_ms(1), eng.kernel_ms(0)

    dt, res, scan_ms = timed_steps(step, fence, 2, steps)
    rows = int(res.kept.sum())
    if dist is not None:
        t = torch.tensor([dt, float(rows)], dtype=torch.float64, device=red_dev)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dt, rows_all = float(tmax[0]), float(t[1])
    else:
        rows_all = float(rows)
    return {"scaling": "strong", "value": rows_all / dt, "unit": "windows/s", "ms_per_step": dt * 1e3,
            "workload": "whole C5 shape: %d scaffolds, %d bases, %d candidate windows, %d rows; rank r scores candidates "
                        "[n r / N, n (r + 1) / N) and counts the k-mers starting in its N-th of the positions; one all-reduce"
                        % (len(lens), sum(lens), n_cand, int(rows_all)),
            "bases": sum(lens), "candidate_windows": n_cand, "rows": int(rows_all), "gbases_per_s": sum(lens) / dt / 1e9,
            "scan_kernel_ms_rank0": scan_ms}


def self_launch(opts, argv):
    """--gpus N without a launcher: become the launcher (no GPU call has been made yet) and relay the ranks' output."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(opts.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--shard-scale", type=float, default=1.0,
                    help="scale every scaffold length of the workload (testing only; 1.0 = the named workload)")
    ap.add_argument("--workload", choices=("c5", "shard"), default="c5",
                    help="c5: the whole C5 shape per GPU (the headline); shard: one 1/8 LPT shard of it per GPU (what rounds 1-3 "
                         "reported; profiling runs)")
    ap.add_argument("--cpu-windows", type=int, default=150, help="windows in the CPU-baseline sample, ~0.1 s each (0 = skip)")
    ap.add_argument("--no-upload", action="store_true", help="skip the `inclusive` block (H2D-inclusive jobs from host memory)")
    ap.add_argument("--repeats", type=float, default=0.0,
                    help="simple repeats per kb in the synthetic shard (profiling runs of the `realistic` shape: 0.35; the headline metric is 0)")
    ap.add_argument("--no-extra", action="store_true", help="skip the cold / strong / realistic blocks (profiling runs)")
    ap.add_argument("--dry-run", action="store_true",
                    help="rendezvous only (gloo, no GPU): proves the N-rank launch path; rank 0 prints {dry_run, world}")
    opts = ap.parse_args(argv)

    if opts.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(opts, argv))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != opts.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (opts.gpus, world, opts.gpus))
    import torch
    dist = None
    if opts.dry_run:
        if world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo", rank=rank, world_size=world)
            t = torch.ones(1)
            dist.all_reduce(t)
            dist.barrier()
            seen = int(t.item())
            dist.destroy_process_group()
        else:
            seen = 1
        if rank == 0:
            print(json.dumps({"dry_run": True, "world": world, "ranks_seen": seen}), flush=True)
        return
    # FRISK_BENCH_REHEARSAL=1: the N-rank code path on ONE GPU (every rank on device 0, gloo instead of RCCL - RCCL refuses two
    # ranks on one device): a functional rehearsal of what the driver runs on N GPUs, its numbers mean nothing
    rehearsal = os.environ.get("FRISK_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    red_dev = "cpu" if rehearsal else "cuda"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from frisk_amd import Engine, synth
    base_lens = c5_lens() if opts.workload == "c5" else synth.c5_shard_lens(8, rank % 8)
    lens = [max(1, int(x * opts.shard_scale)) for x in base_lens]
    eng = Engine(KMIN, KMAX, device=local_rank)
    eng.synth(lens, seed=0xC5 + rank, island_frac=0.02, n_frac=0.07, lower_frac=0.0, repeats_per_kb=opts.repeats)
    n_cand = eng.scan_plan(W, INC)
    total_bases = sum(lens)

    def step():
        eng.profile_reset()
        eng.profile_add(mask_host=False)
        eng.profile_allreduce()
        eng.profile_finalize()
        res = eng.scan(W, INC, pinned=True)                 # returns when the rows are in host memory
        # (kernel timers are read AFTER the step's work is queued: reading the profile kernel's events earlier would park
        #  the host on them and delay the launches behind)
        return res, eng.kernel_ms(1), eng.kernel_ms(0)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(opts.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    scan_ms, prof_ms, rows = [], [], 0
    for _ in range(opts.steps):
        res, tp, ts = step()
        scan_ms.append(ts)
        prof_ms.append(tp)
    fence()
    elapsed = time.perf_counter() - t0
    rows = int(res.kept.sum())                              # (every step emits the same rows: counted once, outside the timed region)
    if dist is not None:
        t = torch.tensor([elapsed, float(rows), float(total_bases), float(n_cand)], dtype=torch.float64, device=red_dev)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
        rows_all, bases_all, cand_all = float(t[1]), float(t[2]), float(t[3])
    else:
        rows_all, bases_all, cand_all = float(rows), float(total_bases), float(n_cand)

    shard_kw = dict(seed=0xC5 + rank, island_frac=0.02, n_frac=0.07, lower_frac=0.0, repeats_per_kb=opts.repeats)
    cold = strong = shard = None
    realistic = []
    if not opts.no_extra:
        few = max(2, min(opts.steps, 10))
        cold_ms, cold_scan_ms, _ = cold_step(eng, (lens, shard_kw), step, fence)
        cold = {"cold_first_step_ms": cold_ms, "cold_first_scan_kernel_ms": cold_scan_ms,
                "note": "first step on a batch that has just become resident: includes the sample (one round of the launch's workgroups) of "
                        "the adaptive counter width; the steps behind `value` reuse the sample's verdict"}
        if world > 1:
            strong = strong_block(eng, dist, torch, rank, world, red_dev, fence, few)
        if world == 1 and opts.shard_scale == 1.0:
            shard, slens = shard_block(eng, rank, lambda: step, fence, few)
            for label, kw in (("unmasked assembly with simple repeats of period 1, 2, 4", synth.REPEATS_UNMASKED),
                              ("unmasked assembly with simple repeats of period 1-6 and 3 % satellite arrays", synth.REPEATS_MIXED),
                              ("soft-masked assembly with simple repeats", synth.REPEATS_SOFT)):
                realistic.append(shape_block(eng, slens, dict(kw, seed=0xC5 + rank), step, fence, few, label))
        eng.synth(lens, **shard_kw)         # the headline batch again (inclusive block, CPU baselines)
        step()
        fence()
    inclusive = None
    if world == 1 and not opts.no_upload:
        inclusive = inclusive_block(eng, lens, step, fence, max(2, min(opts.steps, 6)), rows, total_bases)

    if rank == 0:
        ms_per_step = elapsed * 1e3 / opts.steps
        scan_avg = sum(scan_ms) / len(scan_ms)
        # algorithmic bytes of one scan launch: 2-bit bases once, 40 B per emitted row, genome table once
        b_alg = 0.25 * total_bases + 40.0 * rows + 8.0 * sum(4 ** x for x in range(KMIN, KMAX + 1))
        achieved = b_alg / (scan_avg * 1e-3) / 1e9
        width, handed8, handed16, row_segments = eng.scan_stat()
        side_table = eng.scan_side()
        traffic, binding, pmc_src = None, None, None
        # HBM traffic and issue counters: rocprofv3 --pmc passes of THIS workload, collected offline (separate runs, never
        # combined with tracing) and committed; valid only for the same shard and the same kernel
        tpath = os.path.join(ROOT, "profiles", "r4_pmc_bench.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            same_work = tj.get("workload_bases_per_gpu") == total_bases and tj.get("candidate_windows_per_gpu") == n_cand
            # ... and only for the SOURCES the profile was taken from: the file carries their hash and the profiled kernel's name
            if same_work and tj.get("csrc_sha256") == csrc_hash():
                pmc_src = ("profiles/r4_pmc_bench.json (offline rocprofv3 --pmc passes of this workload and of these sources - "
                           "csrc_sha256 matches; kernel %s; not measured in this run)" % tj.get("kernel"))
                traffic = tj.get("hbm_bytes_per_launch")
                binding = tj.get("binding")
            else:
                pmc_src = ("profiles/r4_pmc_bench.json was taken from other sources or another workload (csrc_sha256 / sizes "
                           "differ): traffic and binding not quoted")
        out = {
            "metric": "windows/sec (k=1..8, w=5kb, s=1kb)", "value": rows_all / (elapsed / opts.steps),
            "unit": "windows/s", "n_gpus": world, "steps": opts.steps, "warmup": opts.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/u4 counts + f64 scores", "data": "synthetic",
            "config": {"workload": ("C5: the whole GRCh38-like synthetic assembly (24 chromosome-scale + 400 small scaffolds, 3.29 Gb, 7% N) "
                                    "on every GPU" if opts.workload == "c5" else
                                    "one 1/8 LPT shard of the C5 shape (GRCh38-like synthetic, 7% N) per GPU") +
                                   ", k=1..8 w=5000 i=1000; step = genome profile + all-reduce + window scan, inputs packed and resident in HBM",
                       "bases_per_gpu": total_bases, "candidate_windows_per_gpu": n_cand, "rows_per_gpu_rank0": rows,
                       "shard_scale": opts.shard_scale, **({"simple_repeats_per_kb": opts.repeats} if opts.repeats else {})},
            "gbases_per_s": bases_all / (elapsed / opts.steps) / 1e9,
            "windowed_gbases_per_s": rows_all * W / (elapsed / opts.steps) / 1e9,      # rows x w: bases looked at, overlap counted
            "scan_kernel_ms": scan_avg, "profile_kernel_ms": sum(prof_ms) / len(prof_ms),
            # (the timed steps one by one: the GPU's clocks are still rising through the first steps behind the input generator's
            #  small launches - the first scan of a fresh batch is ~1.2 ms slower than the fourth even without the sample: the GPU leaving an idle power state, NOTES.md round 4)
            "scan_kernel_ms_first_min_last": [scan_ms[0], min(scan_ms), scan_ms[-1]],
            "scan_kernel_windows_per_s": n_cand / (scan_avg * 1e-3),
            "scan_counter_width": {"bulk_bits": width, "side_table_for_period4_maxmers": side_table, "windows_handed_to_8bit": handed8,
                                   "windows_handed_to_16bit": handed16},
            "scan_row_segments": row_segments,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": pmc_src,
                         "traffic_note": "bytes at the L2's fabric side per scan (2 x FETCH_SIZE + WRITE_SIZE; Infinity Cache hits included).  Most "
                                         "of it is the per-workgroup ring that carries genome-side values from window to window "
                                         "(61 MB in all: beyond L2, inside the 256 MB Infinity Cache) - traded for a 3.5 x cut of the L2 -> L1 line "
                                         "traffic that bound the kernel (`l2_gather`)",
                         "kernel": "scan8_kernel: the two bulk launches of a scan (about 15/16 of the rows, and the rest - whole rounds of the launch's workgroups - on a second stream while the first rows go to the host) + its hand-over launches and finish_rows_kernel: scan_kernel_ms spans them",
                         "algorithmic_bytes_per_launch": b_alg,
                         "bulk_launches_per_scan": row_segments,
                         "duration_ms": scan_avg,
                         "duration_note": "achieved = algorithmic bytes of ONE SCAN / HIP-event time over all of its launches; "
                                          "rocprofv3 --stats lists the bulk kernel with two launches per scan (the bulk of the rows, then "
                                          "a tail of whole rounds of workgroups on a second stream): its 'average' there is their mean, the scan is their sum",
                         "note": "formal bound only: the path is not HBM-limited at any plausible rate (290 B/window, "
                                 "HBM-bound ceiling 2.7e10 windows/s); what binds is in `binding` and `l2_gather`"},
            # what did bind until round 3 (NOTES.md 3.3a): every scored position reads 8 bytes of the 512 KB genome table at a random
            # place = one 128-byte line from L2.  Modelled line traffic of one scan, before and with the ring (a position's value is
            # gathered once per chunk of 16 windows and otherwise read coalesced from the workgroup's ring)
            "l2_gather": (lambda pos, chunk: {
                "lines_are": "128 B per gathered double (vector L1: 32 KB against a 512 KB table)",
                "gather_GB_per_scan_without_ring": rows * pos * 128 / 1e9,
                "time_at_l2_peak_ms_without_ring": rows * pos * 128 / 34.5e12 * 1e3,
                "l2_peak_TBps": 34.5,
                "gather_GB_per_scan_with_ring": rows * (pos / chunk + (INC + 3 * 20) * (chunk - 1) / chunk) * 128 / 1e9
                                                + rows * pos * 8 * 2 / 1e9,
                "note": "with the ring: one window in %d gathers every position, the others the entering range (inc positions, rounded "
                        "up to the lanes that hold them) and read / park the rest as coalesced doubles" % chunk})(W - KMAX + 1, 16),
            "binding": binding,
            "allreduce_path": getattr(eng, "allreduce_path", None) if world > 1 else "none (one rank: no collective)",
            **({"rehearsal_on_one_gpu": True} if rehearsal else {}),
            "inclusive": inclusive,
            "cold": cold,
            "shard": shard,
            "strong": strong,
            "realistic": realistic,
        }
        if opts.cpu_windows > 0:
            cb, worst, np_line, c_line = cpu_baseline(eng, lens[0], opts.cpu_windows)
            model, logical, usable = host_cpu()
            cb["host_cpu"] = "%s, %d logical CPUs (%d usable by this process)" % (model, logical, usable)
            out["cpu_baseline"] = cb
            out["cpu_baseline_numpy"] = np_line
            out["cpu_baseline_c"] = c_line
            out["cpu_sample_max_abs_dKLD"] = worst
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

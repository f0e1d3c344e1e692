#!/usr/bin/env python3
"""Copy the judged summaries of one tools/profile_round.sh directory into profiles/ (tracked):
usage: collect_profiles.py gpurun_out/prof_<tag> r2_<tag>"""
import csv
import json
import os
import shutil
import sys

src, name = sys.argv[1], sys.argv[2]
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
for f, g in (("kernel_stats.csv", "kernel_stats.csv"), ("pmc_summary.json", "pmc_summary.json"), ("bench_line.json", "bench_line.json"),
             ("bench_line_under_rocprof.json", "bench_line_under_rocprof.json"), ("pmc_bench.json", "pmc_bench.json")):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, "%s_%s" % (name, g)))
# the hot kernels' rows of the kernel trace (grid, LDS, registers per dispatch)
kt = os.path.join(src, "kernel_trace.csv")
if os.path.exists(kt):
    rows = list(csv.DictReader(open(kt)))
    hot = [r for r in rows if any(k in r["Kernel_Name"] for k in ("scan8_kernel", "scan_kernel", "profile_add_kernel", "finish_rows"))]
    with open(os.path.join(dst, name + "_kernel_trace_hot.csv"), "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(hot[-24:])
# streamed upload (tools/stream_job.py under rocprofv3 --kernel-trace --memory-copy-trace): the H2D pieces of ONE job against
# the kernels that follow them - expand_runs, one profile_add_kernel per piece, finalize, the scan
mc, uk = os.path.join(src, "upload_memory_copy_trace.csv"), os.path.join(src, "upload_kernel_trace.csv")
if os.path.exists(mc) and os.path.exists(uk):
    copies = [r for r in csv.DictReader(open(mc)) if "HOST_TO_DEVICE" in r["Direction"]]
    kern = [r for r in csv.DictReader(open(uk)) if any(k in r["Kernel_Name"] for k in ("scan8_kernel", "profile_add", "expand_runs", "genome_ivom", "finish_rows"))]
    big = [r for r in copies if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 100000]
    # the last job: from the first big copy behind the last-but-one scan to the end of the last scan
    scans = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in kern if "scan8_kernel" in r["Kernel_Name"]
                   and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 5000000)
    if len(scans) >= 2 and big:
        lo = scans[-2][1]
        t0 = min(int(r["Start_Timestamp"]) for r in big if int(r["Start_Timestamp"]) > lo)
        ev = [(int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, "H2D copy (copy stream; the big ones: 64 MB pieces of the 2-bit codes)")
              for r in copies if int(r["Start_Timestamp"]) >= t0 and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 20000]
        ev += [(int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, r["Kernel_Name"].split("(")[0][:60]) for r in kern
               if int(r["Start_Timestamp"]) >= t0 and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 20000]
        ev.sort()
        under = sum(1 for a, b, w in ev if w.startswith("void profile_add") or w.startswith("profile_add"))
        with open(os.path.join(dst, name + "_stream_overlap.txt"), "w") as fh:
            fh.write("# rocprofv3 --kernel-trace --memory-copy-trace -- python3 tools/stream_job.py 3   (one whole C5-shaped job, 0.25 B/base form)\n"
                     "# the LAST job of the run, every event >= 0.02 ms, t = 0 at its first big H2D copy: the codes cross PCIe in 64 MB pieces\n"
                     "# (copy stream) while one profile_add_kernel per piece runs on the compute stream behind its piece's event; only the\n"
                     "# last piece's kernel, finalize and the scan follow the upload.  %d profile_add launches in this job.\n"
                     "#   start [ms]   duration [ms]   what\n" % under)
            for a, b, what in ev:
                fh.write("%12.3f   %10.3f   %s\n" % (a / 1e6, (b - a) / 1e6, what))
    if os.path.exists(os.path.join(src, "stream_job.json")):
        shutil.copy(os.path.join(src, "stream_job.json"), os.path.join(dst, name + "_stream_job.json"))
print("copied to", dst)

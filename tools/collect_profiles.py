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
# staged upload: H2D copies against the kernels, last steps of the run (rocprofv3 --kernel-trace --memory-copy-trace)
mc, uk = os.path.join(src, "upload_memory_copy_trace.csv"), os.path.join(src, "upload_kernel_trace.csv")
if os.path.exists(mc) and os.path.exists(uk):
    copies = [r for r in csv.DictReader(open(mc)) if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 200000]
    kern = [r for r in csv.DictReader(open(uk)) if any(k in r["Kernel_Name"] for k in ("scan8_kernel", "profile_add", "pack_kernel"))
            and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 100000]
    t0 = min(int(r["Start_Timestamp"]) for r in copies + kern)
    ev = [(int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, "H2D copy (stream %s)" % r["Stream_Id"]) for r in copies if "HOST_TO_DEVICE" in r["Direction"]]
    ev += [(int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, r["Kernel_Name"].split("(")[0][:48]) for r in kern]
    ev.sort()
    with open(os.path.join(dst, name + "_upload_overlap.txt"), "w") as fh:
        fh.write("# rocprofv3 --kernel-trace --memory-copy-trace -- python3 bench.py --steps 3 --warmup 1 --cpu-windows 0\n"
                 "# last 30 events >= 0.1 ms: the `pipelined_ascii` phase - frisk_seq_stage's H2D copies (410 MB per step, copy stream)\n"
                 "# run WHILE profile_add / scan8_kernel of the resident batch run (compute stream); pack_kernel follows the copies.\n"
                 "#   start [ms]   duration [ms]   what\n")
        for a, b, what in ev[-30:]:
            fh.write("%12.3f   %10.3f   %s\n" % (a / 1e6, (b - a) / 1e6, what))
print("copied to", dst)

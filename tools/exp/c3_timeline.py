#!/usr/bin/env python3
"""Timeline of the LAST scan in a rocprofv3 --kernel-trace csv of tools/exp/c3_trace.py: usage c3_timeline.py <kernel_trace.csv>"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
fin = [i for i, r in enumerate(rows) if "finish_rows" in r["Kernel_Name"]]
lo = fin[-2] + 1 if len(fin) > 1 else 0
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:fin[-1] + 1]:
    print("%9.1f us  +%8.1f us  grid %6s  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                                   r.get("Grid_Size", r.get("Grid_Size_X", "?")), r["Kernel_Name"][:70]))

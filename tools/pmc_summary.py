#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 --pmc passes: reads every *counter_collection.csv under <dir>, groups rows by
(kernel name, counter) and prints {kernel: {counter: {launches, mean}}} as JSON.  Values are summed over the
dimensions (XCDs / SEs / instances) of one dispatch first, as rocprofv3 reports one row per dimension."""
import collections
import csv
import glob
import json
import os
import sys


def main(root):
    per = collections.defaultdict(lambda: collections.defaultdict(float))      # (kernel, counter) -> dispatch -> sum
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                kernel = row.get("Kernel_Name") or row.get("kernel_name")
                counter = row.get("Counter_Name") or row.get("counter_name")
                value = float(row.get("Counter_Value") or row.get("counter_value") or 0.0)
                disp = (path, row.get("Dispatch_Id") or row.get("dispatch_id"))
                per[(kernel.split("(")[0], counter)][disp] += value
    out = collections.defaultdict(dict)
    for (kernel, counter), d in sorted(per.items()):
        vals = list(d.values())
        out[kernel][counter] = {"launches": len(vals), "mean": sum(vals) / len(vals)}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else ".")

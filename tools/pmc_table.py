#!/usr/bin/env python3
"""Per-kernel table of raw rocprofv3 PMC counters (one or several --pmc passes of the same command).
    python3 tools/pmc_table.py out.txt dir1 [dir2 ...]      # each dir holds *_counter_collection.csv (+ *_kernel_trace.csv)
Rows: kernel instantiation (+ grid size, so the layers of one instantiation stay apart); columns: launches, average
duration in us (from the kernel trace of the same pass when present), then every counter averaged per launch."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    n = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*$", "", n)


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    agg = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    counters = []
    for d in dirs:
        dur = {}
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                dur[int(r["Dispatch_Id"])] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        seen = set()
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                key = "%s grid=%s" % (short(r["Kernel_Name"]), r.get("Grid_Size", "?"))
                c = r["Counter_Name"]
                if c not in counters:
                    counters.append(c)
                agg[key][c] += float(r["Counter_Value"])
                cnt[key][c] += 1
                did = int(r["Dispatch_Id"])
                if (key, did) not in seen and did in dur:
                    seen.add((key, did))
                    agg[key]["_ns"] += dur[did]
                    cnt[key]["_ns"] += 1
    keys = sorted(agg, key=lambda k: -(agg[k]["_ns"] / max(1, cnt[k]["_ns"])) * max(1, cnt[k]["_ns"]))
    with open(out, "w") as o:
        o.write("# counters averaged per launch; us = kernel-trace duration of the profiled pass\n")
        for k in keys:
            n = max(cnt[k].values())
            us = agg[k]["_ns"] / max(1, cnt[k]["_ns"]) / 1e3
            o.write("%s  launches=%d us=%.1f\n" % (k, n, us))
            o.write("    " + "  ".join("%s=%.4g" % (c, agg[k][c] / max(1, cnt[k][c])) for c in counters if cnt[k][c]) + "\n")
    print(open(out).read()[:6000])


if __name__ == "__main__":
    main()

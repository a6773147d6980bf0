#!/usr/bin/env python3
"""HBM traffic per kernel launch from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected SEPARATELY, as
MI355X_MICROARCH.md 'HBM / rocprofv3 PMC slots' prescribes) of `python3 tools/layer_report.py`:

    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_fetch -o fetch --output-format csv -- python3 tools/layer_report.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_write -o write --output-format csv -- python3 tools/layer_report.py
    python3 tools/pmc_traffic.py gpurun_out/pmc_fetch/fetch_counter_collection.csv gpurun_out/pmc_write/write_counter_collection.csv profiles/r03_pmc_traffic_6890v_b64_f32 6890v_b64_f32
(layer_report.py takes `64 tests/golden/template6890.npz bf16` for the bf16 path -> profiles/r02_pmc_traffic_bf16).  Run it ON THE
BOX that took the passes: the result is stamped with the hash of the kernel sources the library was built from ("_meta"), and
bench.py only quotes it as `roofline.traffic` while that hash matches the library it runs.

Units / corrections (same guide): both counters are in KiB; on gfx950 FETCH_SIZE tallies the 128-byte requests of wide
coalesced reads at 64 bytes, so it is doubled; WRITE_SIZE is exact.  Writes <out>.json (kernel name -> mean bytes per
launch, keyed like rocprofv3 --stats prints the name minus the 'void (anonymous namespace)::' prefix and the
argument list) and <out>.txt (one line per launch of one training step)."""
import csv
import hashlib
import json
import os
import re
import sys
from collections import OrderedDict, defaultdict


def short(name):
    n = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*$", "", n)


def read(path, counter):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                rows.append((int(r["Dispatch_Id"]), short(r["Kernel_Name"]), int(r["Grid_Size"]), float(r["Counter_Value"])))
    rows.sort()
    return rows


def main():
    fetch, write, out = read(sys.argv[1], "FETCH_SIZE"), read(sys.argv[2], "WRITE_SIZE"), sys.argv[3]
    workload = sys.argv[4] if len(sys.argv) > 4 else None       # bench.workload_tag(...) of the profiled run, e.g. 6890v_b64_f32
    # every kernel of libsh_kernels.so, whatever its family (round 4's allow-list of name fragments had no entry for the plane
    # convs, the activation backward, the loss and the image conversion: the planes3 profiles silently lacked them) - i.e.
    # everything that is not torch's (at::), a BLAS library's (Cijk_) or RCCL's
    ours = lambda n: not any(k in n for k in ("at::", "Cijk_", "ccl", "rocprim", "hipcub", "__amd_rocclr"))  # noqa: E731
    agg = defaultdict(lambda: [0, 0.0, 0.0])
    for (_, n, _, v) in fetch:
        if ours(n):
            agg[n][0] += 1; agg[n][1] += 2.0 * v * 1024.0
    cnt_w = defaultdict(int)
    for (_, n, _, v) in write:
        if ours(n):
            cnt_w[n] += 1; agg[n][2] += v * 1024.0
    res = OrderedDict()
    for n, (c, fb, wb) in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
        res[n] = {"launches_profiled": c, "fetch_bytes_per_launch": fb / c, "write_bytes_per_launch": wb / max(1, cnt_w[n]),
                  "hbm_bytes_per_launch": fb / c + wb / max(1, cnt_w[n])}
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from semantichuman_amd import _lib                     # the LOADED library's own identity (sh_build_id) + the switches in force
    res["_meta"] = {"lib_sha16": _lib.build_id(), "workload": workload, "env": _lib.env_overrides(),
                    "corrections": "FETCH_SIZE x 2 (gfx950 tallies 128-byte requests at 64 bytes), WRITE_SIZE exact; KiB units"}
    json.dump(res, open(out + ".json", "w"), indent=1)
    with open(out + ".txt", "w") as f:
        f.write("# HBM bytes per launch, mean over the profiled launches (FETCH_SIZE x2 correction applied; see tools/pmc_traffic.py)\n")
        f.write("%-60s %8s %12s %12s\n" % ("kernel", "launches", "fetch MB", "write MB"))
        for n, r in res.items():
            if n == "_meta":
                continue
            f.write("%-60s %8d %12.1f %12.1f\n" % (n, r["launches_profiled"], r["fetch_bytes_per_launch"] / 1e6, r["write_bytes_per_launch"] / 1e6))
    print(open(out + ".txt").read())


if __name__ == "__main__":
    main()

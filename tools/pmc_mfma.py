#!/usr/bin/env python3
"""MFMA-pipe utilisation and wave-time breakdown per kernel from one rocprofv3 PMC pass of `python3 tools/layer_report.py`:

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \\
        GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/pmc_sq -o sq --output-format csv -- python3 tools/layer_report.py
    python3 tools/pmc_mfma.py gpurun_out/pmc_sq/sq_counter_collection.csv gpurun_out/pmc_sq/sq_kernel_trace.csv profiles/r01_pmc_mfma.txt

Reading the counters (checked against the algorithmic FLOPs of the conv layers: SQ_VALU_MFMA_BUSY_CYCLES x 64 FLOP equals
2*R*B*K*N to the padding): MFMA busy cycles are summed over the 1024 SIMDs, so  busy * 64 FLOP / duration  is the rate at
which the matrix pipes actually worked (padded MFMAs included) and its ratio to 64 FLOP x 1024 SIMDs x CLOCK the fraction
of the launch they were busy.  CLOCK = 2.43 GHz, what tools/clock_probe.py measures under this training load
(profiles/r01_clock_probe.json; GRBM_GUI_ACTIVE brackets more than the kernel and is not used).  SQ_WAVE_CYCLES,
SQ_WAIT_ANY (s_waitcnt / barrier), SQ_WAIT_INST_ANY (issue stall: the matrix pipe still busy with an earlier MFMA) and
SQ_ACTIVE_INST_ANY are in units of 4 cycles, summed over waves; they are given as fractions of the wave lifetime;
the rest of a launch is ramp-up and tail."""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    n = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*$", "", n)


def main():
    cc, kt, out = sys.argv[1:4]
    rows = defaultdict(dict)
    with open(cc) as f:
        for r in csv.DictReader(f):
            d = rows[int(r["Dispatch_Id"])]
            d[r["Counter_Name"]] = float(r["Counter_Value"])
            d["name"] = short(r["Kernel_Name"])
    with open(kt) as f:
        for r in csv.DictReader(f):
            if int(r["Dispatch_Id"]) in rows:
                rows[int(r["Dispatch_Id"])]["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg = defaultdict(lambda: defaultdict(float))
    for d in rows.values():
        if d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) <= 0 or "ns" not in d:
            continue
        a = agg[d["name"]]
        a["n"] += 1
        for k in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "ns"):
            a[k] += d.get(k, 0.0)
    CLOCK = 2.43e9
    lines = ["# per kernel instantiation, summed over the profiled launches of tools/layer_report.py (batch 64, 6890 vertices); see tools/pmc_mfma.py",
             "%-50s %5s %8s %8s %9s | wave lifetime: %8s %10s %8s" % ("kernel", "calls", "avg us", "MFMA TF", "pipe busy", "waitcnt", "pipe stall", "issuing")]
    tot_busy = tot_cap = 0.0
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1]["ns"]):
        busy = a["SQ_VALU_MFMA_BUSY_CYCLES"]
        cap = 1024.0 * a["ns"] * 1e-9 * CLOCK
        wl = max(1.0, a["SQ_WAVE_CYCLES"])
        tot_busy += busy
        tot_cap += cap
        lines.append("%-50s %5d %8.1f %8.1f %8.0f%% | %21.0f%% %9.0f%% %7.0f%%" % (
            n, a["n"], a["ns"] / a["n"] / 1e3, busy * 64.0 / (a["ns"] * 1e-9) / 1e12, 100.0 * busy / cap,
            100.0 * a["SQ_WAIT_ANY"] / wl, 100.0 * a["SQ_WAIT_INST_ANY"] / wl, 100.0 * a["SQ_ACTIVE_INST_ANY"] / wl))
    lines.append("# all MFMA kernels together: matrix pipes busy %.0f%% of their launches' cycles at %.2f GHz" % (100.0 * tot_busy / tot_cap, CLOCK / 1e9))
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()

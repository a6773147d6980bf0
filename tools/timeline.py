"""Timeline of the LAST replayed training step in a rocprofv3 kernel trace (csv): for every kernel its start relative to the step's
first kernel, its duration and the idle gap before it (previous kernel's end -> this start; negative = overlapped with it), then the
totals: busy time, idle time, overlapped time.  Usage: python tools/timeline.py <kernel_trace.csv> [steps_back]"""
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda t: t[0])
    # a step starts with the copy of the batch into the graph's input (the only kernel outside the graph), found by the
    # first-layer kernel's name: gather_gemm_kernel<1, true, false, true, true> (Cin = 3 forward) runs once per step
    marks = [i for i, k in enumerate(ks) if "gather_gemm_kernel<1, true, false, true, true>" in k[2] or "conv_in3" in k[2]]
    if len(marks) < back + 1:                               # the bf16 path: its weight-fragment preparation opens every step
        marks = [i for i, k in enumerate(ks) if "wfrag_prep_kernel" in k[2]]
    if len(marks) < back + 1:
        print("not enough steps in the trace")
        return
    a, b = marks[-back - 1], marks[-back]
    step = ks[a:b]
    t0 = step[0][0]
    print("%-100s %9s %8s %8s" % ("kernel", "start us", "dur us", "gap us"))
    busy_end = step[0][0]
    idle = over = 0.0
    for s, e, n in step:
        gap = (s - busy_end) / 1e3
        if gap > 0:
            idle += gap
        else:
            over += min(e, busy_end) - s if e > s else 0
        print("%-100s %9.1f %8.1f %8.1f" % (n[:100], (s - t0) / 1e3, (e - s) / 1e3, gap))
        busy_end = max(busy_end, e)
    total = (ks[b][0] - t0) / 1e3
    print("step (first kernel -> next step's first kernel): %.1f us; kernels %d; sum of durations %.1f us; idle between kernels %.1f us; "
          "overlapped %.1f us" % (total, len(step), sum(e - s for s, e, _ in step) / 1e3, idle + (ks[b][0] - busy_end) / 1e3, over / 1e3))
    # gaps averaged over all complete steps
    tot_idle, n_steps = 0.0, 0
    for i in range(len(marks) - 1 - 20 if len(marks) > 21 else 0, len(marks) - 1):
        st = ks[marks[i]:marks[i + 1]]
        be = st[0][0]
        g = 0.0
        for s, e, _ in st:
            if s > be:
                g += s - be
            be = max(be, e)
        g += max(0, ks[marks[i + 1]][0] - be)
        tot_idle += g / 1e3
        n_steps += 1
    print("mean idle per step over the last %d steps: %.1f us" % (n_steps, tot_idle / max(1, n_steps)))


if __name__ == "__main__":
    main()

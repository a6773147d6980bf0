#!/bin/bash
# kernel-level profile of the semantic iteration (eager launches so every kernel is visible to rocprofv3)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/sem_prof
rocprofv3 --kernel-trace --stats -d gpurun_out/sem_prof -o sem --output-format csv -- python3 tools/bench_semantic.py --steps 20 > gpurun_out/sem_prof.log 2>&1
f=$(find gpurun_out/sem_prof -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
iters = 30          # 10 warm-up + 20 timed
print("kernels %d, total per iteration %.1f us, launches per iteration %.1f" % (len(rows), tot / iters / 1e3, sum(int(r["Calls"]) for r in rows) / iters))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:60]:
    print("%8.1f us/it %6.1f calls/it %7.2f us avg  %s" % (float(r["TotalDurationNs"]) / iters / 1e3, int(r["Calls"]) / iters, float(r["AverageNs"]) / 1e3, r["Name"][:150]))
PY

#!/bin/bash
# streaming weight gradients: contiguous vertex blocks per wave (SH_WS_VSTRIDE=0) vs vertices dealt round-robin over the row chunks (1)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py tests/test_configs.py tests/test_train_loop.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -2
for v in 0 1 0 1; do
  for m in exact planes3; do
    SH_WS_VSTRIDE=$v python bench.py --f32-mma $m --no-cpu-baseline --no-secondary --steps 100 --warmup 10 2>/dev/null | python -c "import sys,json; r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('vstride=$v $m', round(r['ms_per_step'],4), round(r.get('hip_kernel_ms_per_step'),4), r['roofline']['kernel'][:44], round(r['roofline']['frac'],3))"
  done
done
for v in 0 1; do
  echo "== vstride=$v wgrad launches (exact form)"
  SH_WS_VSTRIDE=$v SH_F32_MMA=exact python tools/layer_report.py 64 2>/dev/null | grep -E "wgrad_stream"
  O=gpurun_out/vs$v; rm -rf $O; mkdir -p $O
  SH_WS_VSTRIDE=$v SH_F32_MMA=exact timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O -o p --output-format csv -- python3 tools/layer_report.py 64 > /dev/null 2>&1
  python3 - "$O" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "wgrad_stream" in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("wgrad_stream_kernel")[1].split("(")[0] + " grid=" + r.get("Grid_Size", "?")
            agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
for k, (v, n) in sorted(agg.items()):
    print("   fetch MB/launch (x2 corrected) %s: %.1f" % (k, 2 * v / n / 1024.0))
PY
  rm -rf $O
done

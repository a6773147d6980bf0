#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05q; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_dataset.py tests/test_optim.py -q -m gpu 2>&1 | tail -3
for d in f32 bf16; do
timeout 900 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --dtype $d > $O/bench_$d.json 2>$O/bench_$d.err
done
timeout 900 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-graph > $O/bench_eager.json 2>$O/bench_eager.err
python - <<'PY'
import json
for f in ("bench_f32","bench_bf16","bench_eager"):
    try:
        d=json.loads([l for l in open("gpurun_out/r05q/%s.json"%f) if l.startswith("{")][-1])
        print(f, round(d["ms_per_step"],4), round(d["value"],1), d["config"]["launch"][:30], d["train_loss_last"], d["recon_l2_mm_after_run"])
    except Exception as e: print(f,"ERR",e)
PY
bash tools/exp/r05_timeline.sh > /dev/null 2>&1; tail -6 gpurun_out/r05tl/timeline.txt | cut -c1-150; head -4 gpurun_out/r05tl/timeline.txt | cut -c1-150

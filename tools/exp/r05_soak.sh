#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05soak; rm -rf $O; mkdir -p $O
for d in f32 bf16; do
timeout 900 python bench.py --steps 4000 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline --dtype $d > $O/soak_$d.json 2>$O/soak_$d.err
timeout 900 python bench.py --steps 4000 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline --dtype $d --no-fused-update > $O/soak_two_$d.json 2>$O/soak_two_$d.err
done
python - <<'PY'
import json
for f in ("soak_f32","soak_two_f32","soak_bf16","soak_two_bf16"):
    try:
        d=json.loads([l for l in open("gpurun_out/r05soak/%s.json"%f) if l.startswith("{")][-1])
        print(f, round(d["ms_per_step"],4), round(d["value"],1), d["train_loss_last"], d["recon_l2_mm_after_run"])
    except Exception as e: print(f,"ERR",e)
PY

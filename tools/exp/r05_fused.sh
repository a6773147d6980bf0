#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05fa; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_optim.py tests/test_bf16.py -q -m gpu -x 2>&1 | tail -5
for d in f32 bf16; do
timeout 900 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --dtype $d > $O/bench_fused_$d.json 2>$O/bench_fused_$d.err
timeout 900 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-fused-update --dtype $d > $O/bench_unfused_$d.json 2>$O/bench_unfused_$d.err
done
python - <<'PY'
import json
for f in ("bench_fused_f32","bench_unfused_f32","bench_fused_bf16","bench_unfused_bf16"):
    try:
        d=json.loads([l for l in open("gpurun_out/r05fa/%s.json"%f) if l.startswith("{")][-1])
        print(f, round(d["ms_per_step"],4), round(d["value"],1), d["config"].get("adam")[:40], d["train_loss_last"], d["recon_l2_mm_after_run"])
        for k in d.get("kernel_breakdown",[])[:4]: print("   ", k["kernel"], round(k["ms_per_step"],4), k.get("frac"))
    except Exception as e: print(f,"ERR",e)
PY

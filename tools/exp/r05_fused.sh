#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05fa; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_optim.py tests/test_headline.py -q -m gpu -x 2>&1 | tail -5
timeout 900 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary > $O/bench_fused.json 2>$O/bench_fused.err
timeout 900 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-fused-update > $O/bench_unfused.json 2>$O/bench_unfused.err
SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 64 > $O/layer.txt 2>/dev/null
python - <<'PY'
import json
for f in ("bench_fused","bench_unfused"):
    try:
        d=json.loads([l for l in open("gpurun_out/r05fa/%s.json"%f) if l.startswith("{")][-1])
        print(f, round(d["ms_per_step"],4), round(d["value"],1), d["config"].get("adam"), d["train_loss_last"], d["recon_l2_mm_after_run"])
        for k in d.get("kernel_breakdown",[])[:8]: print("   ", k["kernel"], round(k["ms_per_step"],4), k.get("frac"))
    except Exception as e: print(f,"ERR",e)
PY
bash tools/exp/r05_timeline.sh > /dev/null 2>&1; grep -n "adam\|linear_bwd_wgt\|^step\|mean idle" gpurun_out/r05tl/timeline.txt | cut -c1-150

#!/bin/bash
# the three data-parallel modes of bench.py in a world of ONE rank on RCCL (what a 1-GPU box can check of the N > 1 path)
cd "$(dirname "$0")/../.."
O=gpurun_out/r05dp; rm -rf $O; mkdir -p $O
SH_BENCH_FORCE_REDUCER=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline > $O/graph.json 2>$O/graph.err; echo "rc=$?"
SH_BENCH_FORCE_REDUCER=1 SH_BENCH_DP_GRAPH=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline > $O/eager.json 2>$O/eager.err; echo "rc=$?"
SH_BENCH_FORCE_REDUCER=1 SH_BENCH_DP_GRAPH=0 SH_BENCH_DP_SAFE=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline > $O/safe.json 2>$O/safe.err; echo "rc=$?"
SH_BENCH_FORCE_REDUCER=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline --shard-optimizer > $O/shard.json 2>$O/shard.err; echo "rc=$?"
SH_BENCH_FORCE_REDUCER=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --dtype bf16 > $O/graph_bf16.json 2>$O/graph_bf16.err; echo "rc=$?"
python - <<'PY'
import json
for f in ("graph","eager","safe","shard","graph_bf16"):
    try:
        d=json.loads([l for l in open("gpurun_out/r05dp/%s.json"%f) if l.startswith("{")][-1])
        print(f, round(d["ms_per_step"],4), d["config"]["launch"][:80], "|", d["config"].get("adam","")[:30], d["train_loss_last"], (d.get("collective") or {}).get("world_size"))
    except Exception as e: print(f,"ERR",e); print(open("gpurun_out/r05dp/%s.err"%f).read()[-1500:])
PY

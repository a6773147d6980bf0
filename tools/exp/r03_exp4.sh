#!/bin/bash
# round 3, experiment 4: skip of no-source entries in backward-data; bench.py as the driver runs it (secondary block);
# the data-parallel step captured into one hipGraph with the RCCL calls inside (world of one rank)
O=gpurun_out/r03e4; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu > $O/tests_all.txt 2>&1
tail -n 4 $O/tests_all.txt
timeout 300 python tools/layer_report.py 64 > $O/lr_skip.txt 2>$O/lr_skip.err
SH_GG_SKIP=0 timeout 300 python tools/layer_report.py 64 > $O/lr_noskip.txt 2>$O/lr_noskip.err
grep -h "total library" $O/lr_*.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_full.json 2>$O/bench_full.err
tail -c 600 $O/bench_full.err
SH_BENCH_FORCE_REDUCER=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/bench_reducer_graph.json 2>$O/bench_reducer_graph.err
SH_BENCH_FORCE_REDUCER=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-graph > $O/bench_reducer_eager.json 2>$O/bench_reducer_eager.err
SH_BENCH_FORCE_REDUCER=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --dtype bf16 > $O/bench_reducer_graph_bf16.json 2>$O/bench_reducer_graph_bf16.err
python - <<'PY'
import json
for f in ("bench_full","bench_reducer_graph","bench_reducer_eager","bench_reducer_graph_bf16"):
    try:
        d=json.loads(open("gpurun_out/r03e4/%s.json"%f).read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d["config"]["launch"], {k:(v.get("ms_per_step") or v.get("ms_per_iteration") or v.get("p50_batch_ms") or v.get("error")) for k,v in d.get("secondary",{}).items()})
    except Exception as e:
        print(f, "ERR", e)
PY

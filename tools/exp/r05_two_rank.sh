#!/bin/bash
# the multi-rank control flow end to end on ONE GPU: two ranks share it, collectives over gloo (host side), every rank under
# its GPU-free supervisor; then the same with an injected rank failure in the first attempt (the retry must produce the line)
cd "$(dirname "$0")/../.."
O=gpurun_out; mkdir -p $O
SH_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/r05_two_rank_gloo.json 2> $O/r05_two_rank_gloo.err
echo "rc=$?"
SH_BENCH_BACKEND=gloo SH_BENCH_TEST_RANK_FAIL=0:1 SH_BENCH_FAIL_GRACE=3 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline > $O/r05_two_rank_retry.json 2> $O/r05_two_rank_retry.err
echo "rc=$?"
# ... and with a rank that HANGS after the rendezvous in the first attempt: the progress watchdog ends it, the next attempt reports
SH_BENCH_BACKEND=gloo SH_BENCH_TEST_RANK_HANG=0:1 SH_BENCH_PHASE_TIMEOUT=20 SH_BENCH_FAIL_GRACE=3 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline > $O/r05_two_rank_hang.json 2> $O/r05_two_rank_hang.err
echo "rc=$?"
python - <<'PY'
import json
for f in ("gpurun_out/r05_two_rank_gloo.json", "gpurun_out/r05_two_rank_retry.json", "gpurun_out/r05_two_rank_hang.json"):
    try:
        r = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f, r["n_gpus"], round(r["ms_per_step"], 3), r["config"]["launch"], r.get("collective", {}).get("world_size"),
              [x["device"] for x in r.get("collective", {}).get("ranks", [])])
    except Exception as e:
        print(f, "ERR", e)
PY
grep -h supervisor $O/r05_two_rank_retry.err $O/r05_two_rank_hang.err | head -8

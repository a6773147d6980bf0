#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05ka; rm -rf $O; mkdir -p $O
run() {  # label, env assignments...
  lbl=$1; shift
  env "$@" timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-secondary --no-roofline > $O/b.json 2>$O/b.err
  python - "$lbl" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open("gpurun_out/r05ka/b.json") if l.startswith("{")][-1])
    print(sys.argv[1], round(d["ms_per_step"],4), round(d["value"],1), d["config"]["launch"][:20])
except Exception as e: print(sys.argv[1], "ERR", e)
PY
}
run default X=1
run packet_capture_0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run packet_capture_1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run hwq_1 GPU_MAX_HW_QUEUES=1
run hwq_2 GPU_MAX_HW_QUEUES=2
run default X=1
run sdma_0 HSA_ENABLE_SDMA=0
run active_wait HIP_LAUNCH_BLOCKING=0 AMD_DIRECT_DISPATCH=1
run default X=1

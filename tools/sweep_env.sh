#!/bin/bash
# A/B tuning knobs of libsh_kernels.so on the GPU box: runs bench.py once per setting and prints
# one compact line each (ms/step + the per-kernel milliseconds measured by the library's HIP events).
# usage: tools/sweep_env.sh "SH_GG_TB=64 SH_WG_TB=32" "SH_GG_TB=16 SH_WG_TB=8" ...
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for setting in "$@"; do
    out=$(env $setting python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1)
    python - "$setting" "$out" <<'EOF'
import json, sys
setting, line = sys.argv[1], sys.argv[2]
try:
    r = json.loads(line)
except Exception as e:
    print(setting, "FAILED", line[:200]); sys.exit(0)
ks = " ".join("%s=%.0fus" % (k["kernel"].replace("_kernel", "").replace(" ", ""), 1e3 * k["ms_per_step"]) for k in r.get("kernel_breakdown", []))
print("%-28s %.3f ms/step  %.0f meshes/s  hip=%.2fms | %s" % (setting, r["ms_per_step"], r["value"], r.get("hip_kernel_ms_per_step", 0), ks))
EOF
done

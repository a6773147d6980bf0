#!/bin/bash
# round 5: evidence for profiles/ (run on the GPU box): tests, bench lines, layer reports, rocprofv3 kernel stats, PMC traffic
O=gpurun_out/r05final; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 2400 python -m pytest tests -q -m gpu -rs > $O/tests_all.txt 2>&1
grep -n "passed\|failed\|FAILED" $O/tests_all.txt | tail -6
timeout 600 python -m pytest tests/test_headline.py tests/test_p3.py -q -m gpu -s -k "full_training or latent_fc" 2>&1 | grep -h "^FC\|template\|passed\|failed" > $O/headline_and_fc_gate.txt
# PMC traffic: FETCH_SIZE and WRITE_SIZE in separate passes, per workload
pmc() {  # tag, env assignment or "-", layer_report args...
  tag=$1; envs=$2; shift 2
  for c in FETCH_SIZE WRITE_SIZE; do
    if [ "$envs" != "-" ]; then export $envs; fi
    timeout 600 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_${c}_$tag -o p --output-format csv -- python3 tools/layer_report.py "$@" > $O/pmc_${c}_$tag.log 2>&1
    if [ "$envs" != "-" ]; then unset ${envs%%=*}; fi
  done
  if [ "$envs" != "-" ]; then export $envs; fi
  python3 tools/pmc_traffic.py $(find $O/pmc_FETCH_SIZE_$tag -name "*counter_collection.csv" | head -1) $(find $O/pmc_WRITE_SIZE_$tag -name "*counter_collection.csv" | head -1) $O/pmc_traffic_$tag $tag > /dev/null 2>$O/pmc_traffic_$tag.err
  if [ "$envs" != "-" ]; then unset ${envs%%=*}; fi
  rm -rf $O/pmc_FETCH_SIZE_$tag $O/pmc_WRITE_SIZE_$tag
}
pmc 6890v_b64_f32 SH_F32_MMA=exact 64
pmc 6890v_b64_f32_planes3 SH_F32_MMA=planes3 64
pmc 6890v_b64_bf16 - 64 tests/golden/template6890.npz bf16
pmc 27554v_b32_f32_planes3 SH_F32_MMA=planes3 32 tests/golden/template27554.npz f32
for w in 6890v_b64_f32 6890v_b64_f32_planes3 6890v_b64_bf16 27554v_b32_f32_planes3; do
  cp $O/pmc_traffic_$w.json profiles/r05_pmc_traffic_$w.json; cp $O/pmc_traffic_$w.txt profiles/r05_pmc_traffic_$w.txt
done
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_f32.json 2>$O/bench_f32.err
timeout 600 python bench.py --steps 20 --warmup 5 --dtype bf16 > $O/bench_bf16.json 2>$O/bench_bf16.err
timeout 600 python bench.py --steps 20 --warmup 5 --f32-mma exact --no-secondary > $O/bench_f32_exact.json 2>$O/bench_f32_exact.err
SH_BENCH_FORCE_REDUCER=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline > $O/bench_reducer_graph_f32.json 2>$O/bench_reducer_graph_f32.err
SH_F32_MMA=exact timeout 300 python tools/layer_report.py 64 > $O/layer_report_f32_exact.txt 2>/dev/null
SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 64 > $O/layer_report_f32_planes3.txt 2>/dev/null
SH_F32_MMA=split3 timeout 300 python tools/layer_report.py 64 > $O/layer_report_f32_split3.txt 2>/dev/null
timeout 300 python tools/layer_report.py 64 tests/golden/template6890.npz bf16 > $O/layer_report_bf16.txt 2>/dev/null
SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 32 tests/golden/template27554.npz f32 > $O/layer_report_config4_f32_planes3.txt 2>/dev/null
SH_F32_MMA=exact timeout 300 python tools/layer_report.py 32 tests/golden/template27554.npz f32 > $O/layer_report_config4_f32_exact.txt 2>/dev/null
SH_F32_MMA=planes3 timeout 300 python tools/layer_report_decode.py > $O/layer_report_decode_planes3.txt 2>/dev/null
timeout 600 python tools/p3_probe.py 64 --both > $O/p3_probe.txt 2>/dev/null
timeout 600 python tools/p3_probe.py 64 --both --adversarial > $O/p3_probe_adversarial.txt 2>/dev/null
SH_F32_MMA=planes3 timeout 600 python tools/bench_decode.py > $O/decode_config5.json 2>/dev/null
# rocprofv3 kernel statistics of the bench command itself
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_f32 -o f32 --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/bench_f32_rocprof_run.json 2>$O/prof_f32.err
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_ex -o ex --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --f32-mma exact > $O/bench_f32_exact_rocprof_run.json 2>$O/prof_ex.err
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_bf16 -o bf16 --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --dtype bf16 > $O/bench_bf16_rocprof_run.json 2>$O/prof_bf16.err
cp $(find $O/prof_f32 -name "*kernel_stats.csv" | head -1) $O/rocprof_kernel_stats_f32_planes3.csv 2>/dev/null
cp $(find $O/prof_ex -name "*kernel_stats.csv" | head -1) $O/rocprof_kernel_stats_f32_exact.csv 2>/dev/null
cp $(find $O/prof_bf16 -name "*kernel_stats.csv" | head -1) $O/rocprof_kernel_stats_bf16.csv 2>/dev/null
rm -rf $O/prof_f32 $O/prof_bf16 $O/prof_ex
# the latent FCs' update inside their weight-gradient kernels: the kernel against the pair it replaces, the step with / without it
timeout 300 python tools/fc_adam_probe.py planes3 2>/dev/null | grep -v amdgpu.ids > $O/fc_adam_probe.txt
timeout 300 python tools/fc_adam_probe.py exact 2>/dev/null | grep -v amdgpu.ids >> $O/fc_adam_probe.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline --no-fused-update > $O/bench_f32_two_kernel_adam.json 2>/dev/null
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline --no-fused-update --dtype bf16 > $O/bench_bf16_two_kernel_adam.json 2>/dev/null
# one replayed step as a timeline (rocprofv3 kernel trace of the bench command)
for d in f32 bf16; do
  timeout 600 rocprofv3 --kernel-trace -d $O/tr_$d -o t --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline --dtype $d > $O/bench_${d}_trace_run.json 2>/dev/null
  python3 tools/timeline.py $(find $O/tr_$d -name "*kernel_trace.csv" | head -1) > $O/timeline_$d.txt
  rm -rf $O/tr_$d
done
ls $O | head -60
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05final/bench_*.json")):
    try:
        d=json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1]); print(f.split("/")[-1], round(d["ms_per_step"],4), d["config"]["launch"][:40], (d.get("roofline") or {}).get("kernel"), (d.get("roofline") or {}).get("frac"), (d.get("roofline") or {}).get("traffic"))
    except Exception as e: print(f,"ERR",e)
PY

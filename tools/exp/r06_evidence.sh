#!/bin/bash
# round 6: evidence for profiles/ (run on the GPU box; copies what README / DESIGN quote into profiles/r06_* itself):
# tests, PMC traffic per workload (FETCH_SIZE / WRITE_SIZE in SEPARATE passes), bench lines, layer reports, rocprofv3 kernel stats,
# timelines, the three-plane weight-gradient probe, the semantic / decode evidence, the trained-model L2
O=gpurun_out/r06final; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
P=profiles
timeout 2400 python -m pytest tests -q -m gpu -rs > $O/tests_all.txt 2>&1
grep -n "passed\|failed\|FAILED" $O/tests_all.txt | tail -6
(grep -E "^=* ?[0-9]+ (passed|failed)" $O/tests_all.txt | tail -2; grep -E "^SKIPPED" $O/tests_all.txt | sed 's/SKIPPED \[[0-9]*\] //' | sort | uniq -c | sort -rn | head -12) > $P/r06_gpu_tests.txt
timeout 900 python -m pytest tests/test_headline.py tests/test_p3.py -q -m gpu -s -k "full_training or latent_fc or replayed" 2>&1 | grep -h "^FC\|template\|passed\|failed" > $P/r06_headline_and_fc_gate.txt
pmc() {  # tag, env assignment or "-", command...
  tag=$1; envs=$2; shift 2
  if [ "$envs" != "-" ]; then export $envs; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_${c}_$tag -o p --output-format csv -- "$@" > $O/pmc_${c}_$tag.log 2>&1
  done
  python3 tools/pmc_traffic.py $(find $O/pmc_FETCH_SIZE_$tag -name "*counter_collection.csv" | head -1) $(find $O/pmc_WRITE_SIZE_$tag -name "*counter_collection.csv" | head -1) $O/pmc_traffic_$tag $tag > /dev/null 2>$O/pmc_traffic_$tag.err
  if [ "$envs" != "-" ]; then unset ${envs%%=*}; fi
  rm -rf $O/pmc_FETCH_SIZE_$tag $O/pmc_WRITE_SIZE_$tag
  cp $O/pmc_traffic_$tag.json $P/r06_pmc_traffic_$tag.json; cp $O/pmc_traffic_$tag.txt $P/r06_pmc_traffic_$tag.txt
}
pmc 6890v_b64_f32 SH_F32_MMA=exact python3 tools/layer_report.py 64
pmc 6890v_b64_f32_planes3 SH_F32_MMA=planes3 python3 tools/layer_report.py 64
pmc 6890v_b64_bf16 - python3 tools/layer_report.py 64 tests/golden/template6890.npz bf16
pmc 27554v_b32_f32_planes3 SH_F32_MMA=planes3 python3 tools/layer_report.py 32 tests/golden/template27554.npz f32
pmc 6890v_b1024_f32_planes3 SH_F32_MMA=planes3 python3 tools/layer_report_decode.py 1024
pmc semantic_6890v_b48_f32_planes3 SH_F32_MMA=planes3 python3 tools/bench_semantic.py --steps 5
# SQ counters: the three-plane weight gradient (probe) and the semantic iteration (VALU instructions of the pair-distance sweep)
A="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU"
B="SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA SQ_WAVES SQ_ACTIVE_INST_SCA"
C="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
i=0; for set in "$A" "$B" "$C"; do i=$((i+1)); timeout 300 rocprofv3 --pmc $set --kernel-trace -d $O/sq_wp3_$i -o p --output-format csv -- python3 tools/wgrad_p3_probe.py 64 --reps=2 > $O/sq_wp3_$i.log 2>&1; done
python3 tools/pmc_table.py $O/sq_wp3_table.txt $O/sq_wp3_1 $O/sq_wp3_2 $O/sq_wp3_3 > /dev/null 2>&1; grep -A1 -E "^wgrad_p3|^wgrad_stream" $O/sq_wp3_table.txt > $P/r06_pmc_sq_wgrad_p3.txt; rm -rf $O/sq_wp3_1 $O/sq_wp3_2 $O/sq_wp3_3
i=0; for set in "$A" "$B"; do i=$((i+1)); SH_F32_MMA=planes3 timeout 300 rocprofv3 --pmc $set --kernel-trace -d $O/sq_sem_$i -o p --output-format csv -- python3 tools/bench_semantic.py --steps 5 > $O/sq_sem_$i.log 2>&1; done
python3 tools/pmc_table.py $O/sq_sem_table.txt $O/sq_sem_1 $O/sq_sem_2 > /dev/null 2>&1; grep -A1 -E "^pairdist|^grouped|^part_|^joint" $O/sq_sem_table.txt > $P/r06_pmc_sq_semantic.txt; rm -rf $O/sq_sem_1 $O/sq_sem_2
# bench lines
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_f32.json 2>$O/bench_f32.err
timeout 600 python bench.py --steps 20 --warmup 5 --dtype bf16 > $O/bench_bf16.json 2>$O/bench_bf16.err
timeout 600 python bench.py --steps 20 --warmup 5 --f32-mma exact --no-secondary > $O/bench_f32_exact.json 2>$O/bench_f32_exact.err
SH_P3_WGRAD=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline > $O/bench_f32_exact_wgrad.json 2>/dev/null
# the same box with round 5's launches (exact weight gradients with their riders, dense transposed tables, one-row lists, every fp32 row written) ...
SH_P3_WGRAD=0 SH_P3_RAGGED=0 SH_P3_GROUPED=0 SH_P3_DROP_FP32=0 SH_P3_YPREV_IMG=0 timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-roofline > $O/bench_f32_round5_launches.json 2>/dev/null
# ... and this round's, measured the same way (100 steps); bf16 with the dense backward-data tables next to the ragged default
timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-roofline > $O/bench_f32_100steps.json 2>/dev/null
SH_BF16_RAGGED=0 timeout 600 python bench.py --steps 100 --warmup 10 --dtype bf16 --no-cpu-baseline --no-secondary --no-roofline > $O/bench_bf16_dense_tables.json 2>/dev/null
timeout 600 python bench.py --steps 100 --warmup 10 --dtype bf16 --no-cpu-baseline --no-secondary --no-roofline > $O/bench_bf16_100steps.json 2>/dev/null
SH_BENCH_FORCE_REDUCER=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline > $O/bench_reducer_graph_f32.json 2>$O/bench_reducer_graph_f32.err
SH_BENCH_FORCE_REDUCER=1 timeout 600 python bench.py --steps 20 --warmup 5 --dtype bf16 --no-cpu-baseline --no-secondary --no-roofline > $O/bench_reducer_graph_bf16_sharded.json 2>$O/bench_reducer_graph_bf16_sharded.err
for f in bench_f32 bench_bf16 bench_f32_exact bench_f32_exact_wgrad bench_f32_round5_launches bench_f32_100steps bench_bf16_dense_tables bench_bf16_100steps bench_reducer_graph_f32 bench_reducer_graph_bf16_sharded; do grep "^{" $O/$f.json | tail -1 > $P/r06_$f.json; done
# layer reports
SH_F32_MMA=exact timeout 300 python tools/layer_report.py 64 > $P/r06_layer_report_f32_exact.txt 2>/dev/null
SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 64 > $P/r06_layer_report_f32_planes3.txt 2>/dev/null
SH_F32_MMA=planes3 SH_P3_WGRAD=0 timeout 300 python tools/layer_report.py 64 > $P/r06_layer_report_f32_planes3_exact_wgrad.txt 2>/dev/null
timeout 300 python tools/layer_report.py 64 tests/golden/template6890.npz bf16 > $P/r06_layer_report_bf16.txt 2>/dev/null
SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 32 tests/golden/template27554.npz f32 > $P/r06_layer_report_config4_f32_planes3.txt 2>/dev/null
SH_F32_MMA=planes3 timeout 300 python tools/layer_report_decode.py > $P/r06_layer_report_decode_planes3.txt 2>/dev/null
SH_F32_MMA=planes3 timeout 300 python tools/bench_semantic.py --steps 30 --graph --layer-report 2>/dev/null | grep -v "^{" > $P/r06_layer_report_semantic.txt
# probes
timeout 300 python tools/wgrad_p3_probe.py 64 --reps=20 2>/dev/null | grep -v amdgpu > $O/wgrad_p3_probe.txt
timeout 300 python tools/wgrad_p3_probe.py 64 --reps=5 --adversarial 2>/dev/null | grep -v amdgpu > $O/wgrad_p3_probe_adversarial.txt
timeout 300 python tools/wgrad_p3_probe.py 32 tests/golden/template27554.npz --reps=10 2>/dev/null | grep -v amdgpu > $O/wgrad_p3_probe_config4.txt
timeout 300 python tools/wgrad_p3_probe.py 48 --reps=10 2>/dev/null | grep -v amdgpu > $O/wgrad_p3_probe_b48.txt
(echo "# tools/wgrad_p3_probe.py: per layer, max|err| / max|ref| against float64 (exact fp32 MFMA kernel / three-plane kernel) for dW and dbias, us (exact incl. its slab reduction / plane kernel alone), slabs"; echo "## 6890 vertices, batch 64, training-scale operands"; cat $O/wgrad_p3_probe.txt; echo "## ... adversarial operands (six decades of dynamic range)"; cat $O/wgrad_p3_probe_adversarial.txt; echo "## 27 554 vertices, spiral 18, batch 32"; cat $O/wgrad_p3_probe_config4.txt; echo "## 6890 vertices, batch 48 (the semantic loop's; 16-row units of different vertices paired)"; cat $O/wgrad_p3_probe_b48.txt) > $P/r06_wgrad_p3_probe.txt
timeout 600 python tools/p3_probe.py 64 --both > $P/r06_p3_probe.txt 2>/dev/null
SH_F32_MMA=planes3 timeout 600 python tools/bench_decode.py > $O/decode_config5.json 2>/dev/null; grep "^{" $O/decode_config5.json | tail -1 > $P/r06_decode_config5.json
# rocprofv3 kernel statistics of the bench command itself
for v in "f32:" "f32_exact:--f32-mma exact" "bf16:--dtype bf16"; do
  n=${v%%:*}; a=${v#*:}
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$n -o p --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary $a > $O/bench_${n}_rocprof_run.json 2>$O/prof_$n.err
  cp $(find $O/prof_$n -name "*kernel_stats.csv" | head -1) $P/r06_rocprof_kernel_stats_$n.csv 2>/dev/null
  grep "^{" $O/bench_${n}_rocprof_run.json | tail -1 > $P/r06_bench_${n}_rocprof_run.json
  rm -rf $O/prof_$n
done
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_sem -o p --output-format csv -- python3 tools/bench_semantic.py --steps 30 --graph > $O/bench_semantic_rocprof_run.json 2>/dev/null
cp $(find $O/prof_sem -name "*kernel_stats.csv" | head -1) $P/r06_rocprof_kernel_stats_semantic.csv 2>/dev/null; rm -rf $O/prof_sem
# one replayed step as a timeline
for d in f32 bf16; do
  timeout 600 rocprofv3 --kernel-trace -d $O/tr_$d -o t --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline --dtype $d > $O/bench_${d}_trace_run.json 2>/dev/null
  python3 tools/timeline.py $(find $O/tr_$d -name "*kernel_trace.csv" | head -1) > $P/r06_timeline_$d.txt
  rm -rf $O/tr_$d
done
# trained-model matched L2
timeout 2000 python tools/trained_l2.py --seeds 5 --eval-every 100 --out $P/r06_trained_l2.json 2>/dev/null | grep -v "Consider\|curve.append" > $P/r06_trained_l2.log.txt
ls $P | grep r06 | head -80
python - <<'PY'
import json,glob
for f in sorted(glob.glob("profiles/r06_bench_*.json")):
    try:
        d=json.loads(open(f).read()); print(f.split("/")[-1], round(d["ms_per_step"],4), d["config"].get("launch","")[:40], (d.get("roofline") or {}).get("kernel"), (d.get("roofline") or {}).get("frac"), (d.get("roofline") or {}).get("traffic"))
    except Exception as e: print(f,"ERR",e)
PY

#!/bin/bash
# gpurun_out/r05final (scratch) -> profiles/r05_* (tracked): the evidence README / DESIGN quote
S=gpurun_out/r05final
for f in bench_f32 bench_bf16 bench_f32_exact bench_f32_rocprof_run bench_f32_exact_rocprof_run bench_bf16_rocprof_run bench_reducer_graph_f32 decode_config5 bench_f32_two_kernel_adam bench_bf16_two_kernel_adam; do
  [ -f $S/$f.json ] && grep '^{' $S/$f.json | tail -1 > profiles/r05_$f.json
done
for f in layer_report_f32_exact layer_report_f32_planes3 layer_report_f32_split3 layer_report_bf16 layer_report_config4_f32_planes3 layer_report_config4_f32_exact layer_report_decode_planes3 p3_probe p3_probe_adversarial headline_and_fc_gate fc_adam_probe timeline_f32 timeline_bf16; do
  [ -f $S/$f.txt ] && grep -v "amdgpu.ids" $S/$f.txt > profiles/r05_$f.txt
done
for k in f32_planes3 f32_exact bf16; do [ -f $S/rocprof_kernel_stats_$k.csv ] && cp $S/rocprof_kernel_stats_$k.csv profiles/r05_rocprof_kernel_stats_$k.csv; done
for w in 6890v_b64_f32 6890v_b64_f32_planes3 6890v_b64_bf16 27554v_b32_f32_planes3; do
  [ -f $S/pmc_traffic_$w.json ] && cp $S/pmc_traffic_$w.json profiles/r05_pmc_traffic_$w.json && cp $S/pmc_traffic_$w.txt profiles/r05_pmc_traffic_$w.txt
done
[ -f $S/tests_all.txt ] && { grep -E "passed|failed" $S/tests_all.txt | tail -1; grep -E "^SKIPPED" $S/tests_all.txt | sed 's/ - no plane-conv.*//' | awk '{c[$3" "$4" "$5" "$6]+=substr($2,2)+0} END {for (k in c) print c[k], k}' | sort -rn | head -5; } > profiles/r05_gpu_tests.txt
ls profiles | grep r05 | wc -l

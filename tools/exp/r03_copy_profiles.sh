#!/bin/bash
# gpurun_out/r03final (scratch) -> profiles/r03_* (tracked): the evidence README / DESIGN quote
S=gpurun_out/r03final
for f in bench_f32 bench_bf16 bench_f32_split3 bench_f32_rocprof_run bench_bf16_rocprof_run bench_f32_split3_rocprof_run bench_reducer_graph_f32 bench_reducer_graph_bf16 decode_config5; do
  grep '^{' $S/$f.json | tail -1 > profiles/r03_$f.json
done
for f in layer_report_f32 layer_report_f32_split3 layer_report_bf16 layer_report_config4_f32 layer_report_decode; do cp $S/$f.txt profiles/r03_$f.txt; done
for k in f32 bf16 f32_split3; do cp $S/rocprof_kernel_stats_$k.csv profiles/r03_rocprof_kernel_stats_$k.csv; done
for w in 6890v_b64_f32 6890v_b64_f32_split3 6890v_b64_bf16 27554v_b32_f32; do cp $S/pmc_traffic_$w.json profiles/r03_pmc_traffic_$w.json; cp $S/pmc_traffic_$w.txt profiles/r03_pmc_traffic_$w.txt; done
grep -E "passed|failed" $S/tests_all.txt | tail -1 > profiles/r03_gpu_tests.txt

#!/bin/bash
# gpurun_out/r04final (scratch) -> profiles/r04_* (tracked): the evidence README / DESIGN quote
S=gpurun_out/r04final
for f in bench_f32 bench_bf16 bench_f32_exact bench_f32_rocprof_run bench_f32_exact_rocprof_run bench_bf16_rocprof_run bench_reducer_graph_f32 bench_reducer_graph_f32_sharded decode_config5 r04_two_rank_gloo r04_two_rank_retry; do
  grep '^{' $S/$f.json | tail -1 > profiles/r04_${f#r04_}.json
done
for f in layer_report_f32_exact layer_report_f32_planes3 layer_report_f32_split3 layer_report_bf16 layer_report_config4_f32_planes3 layer_report_config4_f32_exact p3_probe p3_probe_adversarial; do grep -v "amdgpu.ids" $S/$f.txt > profiles/r04_$f.txt; done
for k in f32_planes3 f32_exact bf16; do cp $S/rocprof_kernel_stats_$k.csv profiles/r04_rocprof_kernel_stats_$k.csv; done
for w in 6890v_b64_f32 6890v_b64_f32_planes3 6890v_b64_bf16 27554v_b32_f32_planes3; do cp $S/pmc_traffic_$w.json profiles/r04_pmc_traffic_$w.json; cp $S/pmc_traffic_$w.txt profiles/r04_pmc_traffic_$w.txt; done
grep -E "passed|failed" $S/tests_all.txt | tail -1 > profiles/r04_gpu_tests.txt
grep -v "amdgpu.ids\|hostname of the client" $S/two_rank_gloo.txt > profiles/r04_two_rank_gloo.txt
cp gpurun_out/r04_pmc_p3/p3_kernels.txt profiles/r04_pmc_p3_kernels.txt
cp gpurun_out/r04_p3_tune.txt profiles/r04_p3_tune.txt; cp gpurun_out/r04_p3_occ.txt profiles/r04_p3_occupancy.txt
ls profiles | grep r04 | wc -l

#!/bin/bash
# round 6: per-kernel evidence for the semantic iteration (3 passes x 16 meshes) and for the decode workload (batch 1024):
# library layer report, rocprofv3 kernel stats, HBM traffic (FETCH_SIZE / WRITE_SIZE in SEPARATE passes)
O=gpurun_out/r06sem; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
export SH_F32_MMA=planes3
timeout 300 python tools/bench_semantic.py --steps 30 --graph --layer-report > $O/layer_report_semantic.txt 2>$O/layer_report_semantic.err
tail -25 $O/layer_report_semantic.txt | cut -c1-200
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_sem -o sem --output-format csv -- python3 tools/bench_semantic.py --steps 30 --graph > $O/bench_semantic_rocprof_run.json 2>$O/prof_sem.err
cp $(find $O/prof_sem -name "*kernel_stats.csv" | head -1) $O/rocprof_kernel_stats_semantic.csv 2>/dev/null; rm -rf $O/prof_sem
head -12 $O/rocprof_kernel_stats_semantic.csv | cut -c1-160
pmc() {  # tag, command...
  tag=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_${c}_$tag -o p --output-format csv -- "$@" > $O/pmc_${c}_$tag.log 2>&1
  done
  python3 tools/pmc_traffic.py $(find $O/pmc_FETCH_SIZE_$tag -name "*counter_collection.csv" | head -1) $(find $O/pmc_WRITE_SIZE_$tag -name "*counter_collection.csv" | head -1) $O/pmc_traffic_$tag $tag > /dev/null 2>$O/pmc_traffic_$tag.err
  rm -rf $O/pmc_FETCH_SIZE_$tag $O/pmc_WRITE_SIZE_$tag
}
pmc 6890v_b1024_f32_planes3 python3 tools/layer_report_decode.py 1024
pmc semantic_6890v_b48_f32_planes3 python3 tools/bench_semantic.py --steps 5
head -14 $O/pmc_traffic_6890v_b1024_f32_planes3.txt; head -14 $O/pmc_traffic_semantic_6890v_b48_f32_planes3.txt
timeout 300 python tools/layer_report_decode.py 1024 > $O/layer_report_decode_planes3.txt 2>/dev/null

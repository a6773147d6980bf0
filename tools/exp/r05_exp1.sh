#!/bin/bash
# round 5, experiment 1: headline-form parity tests; XCD vertex ranges in the streaming weight gradient (A/B + PMC traffic)
O=gpurun_out/r05e1; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 900 python -m pytest tests/test_headline.py -q -m gpu -x -s > $O/test_headline.txt 2>&1; tail -5 $O/test_headline.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_p3.py tests/test_configs.py -q -m gpu -x > $O/test_parity.txt 2>&1; tail -4 $O/test_parity.txt
for x in 0 1; do
  SH_WS_XCD_RANGES=$x SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 64 > $O/layer_planes3_xg$x.txt 2>/dev/null
  SH_WS_XCD_RANGES=$x SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 32 tests/golden/template27554.npz f32 > $O/layer_c4_planes3_xg$x.txt 2>/dev/null
done
grep -h "wgrad\|total" $O/layer_planes3_xg0.txt; echo; grep -h "wgrad\|total" $O/layer_planes3_xg1.txt; echo
grep -h "wgrad\|total" $O/layer_c4_planes3_xg0.txt; echo; grep -h "wgrad\|total" $O/layer_c4_planes3_xg1.txt
pmc() {  # tag, layer_report args...
  tag=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_${c}_$tag -o p --output-format csv -- python3 tools/layer_report.py "$@" > $O/pmc_${c}_$tag.log 2>&1
  done
  python3 tools/pmc_traffic.py $(find $O/pmc_FETCH_SIZE_$tag -name "*counter_collection.csv" | head -1) $(find $O/pmc_WRITE_SIZE_$tag -name "*counter_collection.csv" | head -1) $O/pmc_traffic_$tag $tag > /dev/null 2>$O/pmc_traffic_$tag.err
  rm -rf $O/pmc_FETCH_SIZE_$tag $O/pmc_WRITE_SIZE_$tag
}
export SH_F32_MMA=planes3
export SH_WS_XCD_RANGES=0; pmc xg0_6890 64
export SH_WS_XCD_RANGES=1; pmc xg1_6890 64
pmc xg1_27554 32 tests/golden/template27554.npz f32
grep -h "wgrad" $O/pmc_traffic_xg0_6890.txt; echo; grep -h "wgrad" $O/pmc_traffic_xg1_6890.txt; echo; grep -h "wgrad" $O/pmc_traffic_xg1_27554.txt

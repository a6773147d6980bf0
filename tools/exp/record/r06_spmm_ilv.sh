#!/bin/bash
# round 6: the re-sampling / pre-sum launches with 4 (2) work items of a workgroup in flight together (spmm_quad_u_kernel, SH_SPMM_ILV):
# bitwise check of a whole training step against the one-chain kernels, per-launch times, step time A/B (same box, alternating)
O=gpurun_out/r06ilv; rm -rf $O; mkdir -p $O
for v in 4 2 1; do SH_SPMM_ILV=$v timeout 600 python tools/drop_fp32_check.py > $O/check_$v.txt 2>&1; grep -c DIGEST $O/check_$v.txt; done
cmp <(grep DIGEST $O/check_4.txt) <(grep DIGEST $O/check_1.txt) && cmp <(grep DIGEST $O/check_2.txt) <(grep DIGEST $O/check_1.txt) && echo "DIGESTS EQUAL (ilv 4, 2, 1)"
for v in 4 2 1; do
  SH_SPMM_ILV=$v SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 64 tests/golden/template6890.npz f32 > $O/layer_$v.txt 2>&1
  echo "== ilv $v: $(grep -E '^spmm' $O/layer_$v.txt | awk '{printf "%s ", $(NF)}') | $(grep total $O/layer_$v.txt)"
done
for rep in 1 2; do for v in 4 2 1; do
  SH_SPMM_ILV=$v timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-roofline > $O/bench.json 2>$O/bench.err
  echo "== f32 ilv=$v rep $rep: $(python -c "import json; d=json.loads([l for l in open('$O/bench.json') if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'])" 2>&1 | tail -1)"
done; done
for v in 4 1; do
  SH_SPMM_ILV=$v timeout 400 python bench.py --steps 40 --warmup 10 --batch 32 --template tests/golden/template27554.npz --no-cpu-baseline --no-secondary --no-roofline > $O/bench_c4.json 2>$O/bench_c4.err
  echo "== config 4 ilv=$v: $(python -c "import json; d=json.loads([l for l in open('$O/bench_c4.json') if l.startswith('{')][-1]); print(d['ms_per_step'])" 2>&1 | tail -1)"
  SH_SPMM_ILV=$v SH_F32_MMA=planes3 timeout 300 python tools/bench_decode.py --latents 20480 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('decode ilv=$v: p50 %.4f ms' % d['p50_batch_ms'])"
done
timeout 900 python -m pytest tests/test_p3.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2

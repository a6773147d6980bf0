#!/bin/bash
# PMC passes over tools/wgrad_p3_probe.py: what the three-plane weight-gradient kernel waits for (round 6)
O=gpurun_out/r06_pmc_wp3
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
A="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU"
B="SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA SQ_WAVES SQ_ACTIVE_INST_SCA"
C="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
D="FETCH_SIZE"
E="WRITE_SIZE"
i=0
for set in "$A" "$B" "$C" "$D" "$E"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace -d $O/p$i -o p --output-format csv -- python3 tools/wgrad_p3_probe.py 64 --reps=2 > $O/p$i.log 2>&1
done
python3 tools/pmc_table.py $O/table.txt $O/p1 $O/p2 $O/p3 $O/p4 $O/p5 > /dev/null 2>&1
grep -A1 -E "^wgrad_p3|^wgrad_stream" $O/table.txt > $O/wgrad_kernels.txt
rm -rf $O/p1 $O/p2 $O/p3 $O/p4 $O/p5
cat $O/wgrad_kernels.txt | cut -c1-1200

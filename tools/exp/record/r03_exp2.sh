#!/bin/bash
# round 3, experiment 2: what bounds the conv kernels?  PMC passes over tools/layer_report.py, exact and split3 forms
O=gpurun_out/r03e2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
P1="SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
P2="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P3="TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum"
P4="TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum"
P5="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TD_TD_BUSY_sum"
P6="TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TD_TC_STALL_sum"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/ta_probe tools/exp/ta_probe.hip 2>/dev/null && timeout 120 /tmp/ta_probe > $O/ta_probe.txt 2>&1
cat $O/ta_probe.txt
SH_GG_FILL=1 timeout 300 python tools/layer_report.py 64 > $O/lr_exact_fill1.txt 2>&1
SH_GG_FILL=1 SH_GG_RT=1 timeout 300 python tools/layer_report.py 64 > $O/lr_exact_fill1_rt1.txt 2>&1
for mode in exact; do
  export SH_F32_MMA=$mode
  i=0
  for P in "$P1" "$P2" "$P3" "$P4" "$P5" "$P6"; do
    i=$((i+1))
    timeout 400 rocprofv3 --pmc $P --kernel-trace -d $O/${mode}_p$i -o p --output-format csv -- python3 tools/layer_report.py 64 > $O/${mode}_p$i.log 2>&1
  done
  python3 tools/pmc_table.py $O/pmc_$mode.txt $O/${mode}_p1 $O/${mode}_p2 $O/${mode}_p3 $O/${mode}_p4 $O/${mode}_p5 $O/${mode}_p6 > /dev/null
  rm -rf $O/${mode}_p1 $O/${mode}_p2 $O/${mode}_p3 $O/${mode}_p4 $O/${mode}_p5 $O/${mode}_p6
done
ls -la $O

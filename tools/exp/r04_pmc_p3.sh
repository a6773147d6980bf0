#!/bin/bash
# PMC passes over one planes3 training step (tools/layer_report.py): what the three-plane conv kernels wait for.
cd "$(dirname "$0")/../.."
O=gpurun_out/r04_pmc_p3
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp SH_F32_MMA=planes3
A="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU"
B="SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM"
C="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
D="TD_TD_BUSY_sum TD_TC_STALL_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"
i=0
for set in "$A" "$B" "$C" "$D"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace -d $O/p$i -o p --output-format csv -- python3 tools/layer_report.py 64 > $O/p$i.log 2>&1
done
python3 tools/pmc_table.py $O/table.txt $O/p1 $O/p2 $O/p3 $O/p4 > /dev/null 2>&1
grep -A1 -E "^conv_p3|^spmm_kernel<true, true>|^wgrad_split3_kernel<2>" $O/table.txt > $O/p3_kernels.txt
rm -rf $O/p1 $O/p2 $O/p3 $O/p4
wc -l $O/table.txt

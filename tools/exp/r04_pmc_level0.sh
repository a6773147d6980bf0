#!/bin/bash
# PMC passes over one planes3 training step: what the thin level-0 kernels (3 -> 16 forward, 16 -> 3 forward / weight gradient) wait for.
cd "$(dirname "$0")/../.."
O=gpurun_out/r04_pmc_level0
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp SH_F32_MMA=planes3
A="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU"
B="SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM"
C="SQ_WAVES SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_ANY SQ_INSTS_SMEM"
i=0
for set in "$A" "$B" "$C"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace -d $O/p$i -o p --output-format csv -- python3 tools/layer_report.py 64 > $O/p$i.log 2>&1
done
python3 tools/pmc_table.py $O/table.txt $O/p1 $O/p2 $O/p3 > /dev/null 2>&1
grep -A1 -E "^wgrad_thin|^wgrad_swap|^conv_out3|^wgrad_stream_kernel<1" $O/table.txt > $O/level0_kernels.txt
rm -rf $O/p1 $O/p2 $O/p3
wc -l $O/table.txt

#!/bin/bash
O=gpurun_out/r05e23; rm -rf $O; mkdir -p $O
run() {  # tag env...
  tag=$1; shift
  for rep in 1 2; do
    env "$@" SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 64 > $O/${tag}_$rep.txt 2>&1
  done
  python - "$tag" <<'PY'
import re,sys
tag=sys.argv[1]
def rd(f):
    fam={}
    for l in open(f):
        if l.startswith("total"): fam["total"]=float(l.split()[3]); continue
        m=re.match(r"(\S+?)[<\s].*?\s([\d.]+)(\s+[\d.]+)?\s*$", l)
        if not m or l.startswith("kernel"): continue
        fam[m.group(1)]=fam.get(m.group(1),0)+float(m.group(2))
    return fam
a=rd("gpurun_out/r05e23/%s_1.txt"%tag); b=rd("gpurun_out/r05e23/%s_2.txt"%tag)
print("%-28s"%tag, " ".join("%s %.1f"%(k,(a[k]+b.get(k,a[k]))/2) for k in ("total","wgrad_stream","spmm","linear_fwd_x3","linear_bwd_data_x3","linear_bwd_wgt_x3","conv_p3","conv_p3s") if k in a))
PY
}
run base SH_NOP=1
run tail256 SH_WS_TAIL_BLOCKS=256
run tail512 SH_WS_TAIL_BLOCKS=512
run tail1536 SH_WS_TAIL_BLOCKS=1536
run spmm2048 SH_SPMM_GRID=2048
run spmm8192 SH_SPMM_GRID=8192
run lin1536 SH_LIN_ITEMS=1536
run lin2048 SH_LIN_ITEMS=2048
run lin768 SH_LIN_ITEMS=768
run base2 SH_NOP=1

#!/bin/bash
O=gpurun_out/r05e18; rm -rf $O; mkdir -p $O
ALT=$PWD/semantichuman_amd/lib_alt/libsh_kernels.so
for rep in 1 2; do
  SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 64 > $O/base_$rep.txt 2>&1
  SH_KERNEL_LIB=$ALT SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 64 > $O/alt_$rep.txt 2>&1
done
SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 32 tests/golden/template27554.npz f32 > $O/c4_base.txt 2>&1
SH_KERNEL_LIB=$ALT SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 32 tests/golden/template27554.npz f32 > $O/c4_alt.txt 2>&1
python - <<'PY'
import re
def rd(f):
    out=[]
    for l in open(f):
        m=re.match(r"(\S.*?)\s{2,}(\S.*?)\s+([\d.]+)(\s+[\d.]+)?\s*$", l)
        if l.startswith("total"): out.append(("total","",float(l.split()[3]))); continue
        if not m or l.startswith("kernel"): continue
        out.append((m.group(1), m.group(2)[:46], float(m.group(3))))
    return out
O="gpurun_out/r05e18/"
for tag,new,old in (("6890",["alt_1.txt","alt_2.txt"],["base_1.txt","base_2.txt"]),("27554",["c4_alt.txt"],["c4_base.txt"])):
    n=[rd(O+f) for f in new]; o=[rd(O+f) for f in old]
    print("====",tag)
    for i,(k,sh,_) in enumerate(n[0]):
        a=sum(x[i][2] for x in n)/len(n); b=sum(x[i][2] for x in o)/len(o)
        if k.startswith("conv_p3<") or k=="total": print("%-44s %-46s whole-rounds %7.1f shipped %7.1f  %+5.1f%%"%(k,sh,a,b,100*(a-b)/b))
PY

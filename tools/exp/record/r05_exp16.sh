#!/bin/bash
O=gpurun_out/r05e16; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_headline.py tests/test_configs.py tests/test_p3.py tests/test_bf16.py -q -m gpu -x > $O/tests.txt 2>&1; tail -3 $O/tests.txt | cut -c1-250
ALT=$PWD/semantichuman_amd/lib_alt/libsh_kernels.so
for rep in 1 2; do
  SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 64 > $O/p3_new_$rep.txt 2>&1
  SH_KERNEL_LIB=$ALT SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 64 > $O/p3_old_$rep.txt 2>&1
done
SH_F32_MMA=exact timeout 300 python tools/layer_report.py 64 > $O/ex_new.txt 2>&1
SH_KERNEL_LIB=$ALT SH_F32_MMA=exact timeout 300 python tools/layer_report.py 64 > $O/ex_old.txt 2>&1
timeout 300 python tools/layer_report.py 64 tests/golden/template6890.npz bf16 > $O/bf_new.txt 2>&1
SH_KERNEL_LIB=$ALT timeout 300 python tools/layer_report.py 64 tests/golden/template6890.npz bf16 > $O/bf_old.txt 2>&1
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
O="gpurun_out/r05e16/"
for tag,new,old in (("planes3",["p3_new_1.txt","p3_new_2.txt"],["p3_old_1.txt","p3_old_2.txt"]),("exact",["ex_new.txt"],["ex_old.txt"]),("bf16",["bf_new.txt"],["bf_old.txt"])):
    n=[rd(O+f) for f in new]; o=[rd(O+f) for f in old]
    print("====",tag)
    for i,(k,sh,_) in enumerate(n[0]):
        a=sum(x[i][2] for x in n)/len(n); b=sum(x[i][2] for x in o)/len(o)
        if abs(a-b) > 0.03*b or k=="total": print("%-44s %-46s new %7.1f old %7.1f  %+5.1f%%"%(k,sh,a,b,100*(a-b)/b))
PY

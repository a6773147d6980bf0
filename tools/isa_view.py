#!/usr/bin/env python3
"""Compact view of a kernel's instruction schedule around its MFMA loop (from hipcc -S output):
M=v_mfma G=global_load r=ds_read w=ds_write W:=s_waitcnt BAR=s_barrier br=branch |=label.
usage: tools/isa_view.py file.s <substring of mangled kernel name> [...]"""
import re
import sys


def view(asm, key):
    m = re.search(r'^(_Z\w*' + re.escape(key) + r'\w*):[^\n]*\n(.*?)\.Lfunc_end', asm, re.S | re.M)
    if not m:
        return key + ": not found"
    body = m.group(2).split('\n')
    idx = [i for i, l in enumerate(body) if 'v_mfma' in l]
    if not idx:
        return key + ": no mfma"
    seg = body[max(0, idx[0] - 80):idx[-1] + 30]
    ops = []
    for l in seg:
        t = l.strip().split('//')[0].split(';')[0].strip()
        if not t:
            continue
        op = t.split()[0]
        if op.startswith('s_waitcnt'):
            ops.append(t.replace('s_waitcnt ', 'W:').replace(' ', ''))
        elif op.startswith('v_mfma'):
            ops.append('M')
        elif op.startswith('global_load') or op.startswith('buffer_load'):
            ops.append('G')
        elif op.startswith('ds_read'):
            ops.append('r')
        elif op.startswith('ds_write'):
            ops.append('w')
        elif op.startswith('s_barrier'):
            ops.append('BAR')
        elif op.startswith('s_cbranch') or op.startswith('s_branch'):
            ops.append('br')
        elif t.startswith('.LBB'):
            ops.append('|')
    out, prev, n = [], None, 0
    for o in ops:
        if o == prev:
            n += 1
        else:
            if prev:
                out.append(prev + (str(n) if n > 1 else ''))
            prev, n = o, 1
    out.append(prev + (str(n) if n > 1 else ''))
    return m.group(1)[:90] + "\n  " + ' '.join(out)


if __name__ == "__main__":
    asm = open(sys.argv[1]).read()
    for k in sys.argv[2:]:
        print(view(asm, k))

#!/usr/bin/env python3
"""Instruction mix per basic block of one kernel in hipcc's device assembly (-S --cuda-device-only).
usage: tools/isa_blocks.py file.s kernel-substring [min_insts]
Prints, per block (label .. next label): VALU / SALU / LDS / VMEM / scratch / other counts, whether the block ends in a backward
branch (a loop), and totals for the kernel.  A development aid: what is inside the sample loops, what the state machine costs."""
import re, sys
from collections import Counter

def kind(op):
    if op.startswith('scratch_'): return 'scratch'
    if op.startswith(('global_', 'flat_', 'buffer_')): return 'vmem'
    if op.startswith('ds_'): return 'lds'
    if op.startswith('v_'): return 'valu'
    if op.startswith('s_'): return 'salu'
    return 'other'

def main():
    path, sub = sys.argv[1], sys.argv[2]
    min_insts = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    lines = open(path).read().split('\n')
    start = None
    for i, l in enumerate(lines):
        m = re.match(r'^(_Z\w+):', l)
        if m:
            if start is not None: end = i; break
            if sub in m.group(1): start = i; name = m.group(1)
    else:
        end = len(lines)
    blocks = []; cur = ['entry', Counter(), Counter(), []]
    labels = {}
    for i in range(start + 1, end):
        l = lines[i].split(';')[0].rstrip()
        if not l.strip(): continue
        m = re.match(r'^(\.LBB\w+):', l)
        if m:
            blocks.append(cur); cur = [m.group(1), Counter(), Counter(), []]; labels[m.group(1)] = len(blocks)
            continue
        if l.startswith('\t.') or l.startswith('.'): continue
        op = l.split()[0]
        cur[1][kind(op)] += 1; cur[2][op] += 1
        if op.startswith(('s_cbranch', 's_branch')):
            cur[3].append(l.split()[-1])
    blocks.append(cur)
    tot = Counter(); ops = Counter()
    print(name)
    for bi, (lab, k, o, br) in enumerate(blocks):
        tot.update(k); ops.update(o)
        n = sum(k.values())
        back = [t for t in br if t in labels and labels[t] <= bi]
        if n >= min_insts or back:
            print('%-12s n=%5d valu=%5d salu=%4d lds=%4d vmem=%3d scratch=%3d%s' % (lab, n, k['valu'], k['salu'], k['lds'], k['vmem'], k['scratch'],
                  ('  LOOP->' + ','.join(back)) if back else ''))
    print('total', dict(tot))
    print('top ops', ops.most_common(40))

main()

#!/usr/bin/env python3
"""Instruction mix of one kernel in a hipcc -S listing (static counts; basic blocks listed with their sizes so that a loop body
can be told from the prologue).   python tools/isa_mix.py attn.s attn_bwd_p_kernelILb1ELb1E"""
import collections
import re
import sys


def grp(op):
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('global_', 'buffer_', 'flat_')): return 'vmem'
    if op.startswith('s_waitcnt'): return 'waitcnt'
    if op.startswith(('s_barrier', 's_sleep', 's_nop')): return 'sync'
    if op.startswith('s_'): return 'salu'
    if op.startswith(('v_exp', 'v_rcp', 'v_log', 'v_rsq', 'v_sqrt')): return 'trans'
    if op.startswith('v_accvgpr'): return 'acc-mov'
    if op.startswith('v_'): return 'valu'
    return 'other'


lines = open(sys.argv[1]).read().split('\n')
i0 = next(i for i, l in enumerate(lines) if re.match(r'^_Z\w*' + re.escape(sys.argv[2]) + r'\w*:', l))
i1 = next(i for i in range(i0, len(lines)) if 's_endpgm' in lines[i])
blocks, cur = [], ['entry', collections.Counter(), collections.Counter()]
for l in lines[i0 + 1:i1 + 1]:
    t = l.strip()
    if re.match(r'^\.LBB\w+:', t):
        blocks.append(cur)
        cur = [t.split(':')[0], collections.Counter(), collections.Counter()]
        continue
    if not t or t.startswith(('.', ';', '//')) or t.endswith(':'):
        continue
    op = t.split()[0]
    cur[1][grp(op)] += 1
    cur[2][op] += 1
blocks.append(cur)
tot = collections.Counter()
for b in blocks:
    tot.update(b[1])
print('total', sum(tot.values()), dict(tot))
for name, g, ops in blocks:
    n = sum(g.values())
    if n >= 40:
        print('%-12s %5d  %s' % (name, n, dict(g)))
        if '-v' in sys.argv:
            print('     ', ops.most_common(14))

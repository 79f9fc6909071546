#!/usr/bin/env python3
"""Static instruction mix of the loops of one kernel in a hipcc -S listing:  tools/isa_stats.py file.s <substring of mangled name>
(the MH iteration bodies are straight-line code, so the static count of the biggest loops ~ the dynamic count per iteration)"""
import collections
import re
import sys

src = open(sys.argv[1]).read().split('\n')
want = sys.argv[2]
starts = [i for i, l in enumerate(src) if re.match(r'_Z\w+:', l)]
for si, s0 in enumerate(starts):
    name = src[s0].split(':', 1)[0]
    if want not in name:
        continue
    end = starts[si + 1] if si + 1 < len(starts) else len(src)
    labels, ins = {}, []
    for l in src[s0 + 1:end]:
        l = l.strip()
        m = re.match(r'(\.LBB\d+_\d+):', l)
        if m:
            labels[m.group(1)] = len(ins); continue
        if l.startswith('.Lfunc_end'):
            break
        if not l or l.startswith(';') or l.startswith('.'):
            continue
        ins.append(l)                      # (a kernel may hold several s_endpgm: early exits precede the big bodies)
    print(name, 'total instructions', len(ins))
    loops = []
    for i, l in enumerate(ins):
        m = re.match(r's_cbranch_\w+ (\.LBB\d+_\d+)|s_branch (\.LBB\d+_\d+)', l)
        if m:
            lab = m.group(1) or m.group(2)
            if lab in labels and labels[lab] <= i:
                loops.append((labels[lab], i))
    for a, b in sorted(loops, key=lambda x: x[0] - x[1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 3]:
        c = collections.Counter()
        for l in ins[a:b + 1]:
            op = l.split()[0]
            if op.startswith('scratch_load'): c['scratch_load'] += 1
            elif op.startswith('scratch_store'): c['scratch_store'] += 1
            elif op.startswith('ds_read') or op.startswith('ds_load'): c['ds_read'] += 1
            elif op.startswith('ds_write') or op.startswith('ds_store'): c['ds_write'] += 1
            elif op.startswith('v_accvgpr'): c['accvgpr_mov'] += 1
            elif op.startswith('v_readlane') or op.startswith('v_writelane') or op.startswith('v_readfirstlane'): c['lane_ops'] += 1
            elif op.startswith('s_nop'): c['s_nop'] += 1
            elif op.startswith('v_') and ('_f64' in op): c['f64:' + ('fma' if 'fma' in op else 'mul' if 'mul' in op else 'add' if 'add' in op else 'rcp/sqrt/rsq' if re.search('rcp|sqrt|rsq', op) else 'other')] += 1
            elif op.startswith('v_'): c['valu_other'] += 1
            elif op.startswith('s_load') or op.startswith('s_buffer'): c['smem'] += 1
            elif op.startswith('s_'): c['salu'] += 1
            elif op.startswith('global_'): c['global'] += 1
            else: c['other:' + op] += 1
        valu = sum(v for k, v in c.items() if k.startswith('f64') or k in ('valu_other', 'accvgpr_mov', 'lane_ops'))
        print('  loop [%d..%d] %d instructions, VALU %d:' % (a, b, b - a + 1, valu), dict(sorted(c.items())))

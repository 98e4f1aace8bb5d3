#!/usr/bin/env python3
"""Instruction mix per kernel from a hipcc -S --cuda-device-only listing."""
import re, sys, collections
txt = open(sys.argv[1]).read().splitlines()
pat = sys.argv[2] if len(sys.argv) > 2 else "Li256E"
cur, ops = None, None
for line in txt:
    m = re.match(r'^(_ZN12_GLOBAL__N_1\w+):', line)
    if m:
        cur = m.group(1); ops = collections.Counter(); continue
    if cur and line.startswith('.Lfunc_end'):
        if pat in cur:
            tot = sum(ops.values())
            grp = collections.Counter()
            for k, v in ops.items():
                g = ('v_pk' if k.startswith('v_pk') else 'valu' if k.startswith('v_') else 'salu' if k.startswith('s_') else
                     'lds' if k.startswith('ds_') else 'vmem')
                grp[g] += v
            print(cur[:80], 'total', tot, dict(grp))
            print('   ', [(k, v) for k, v in ops.most_common(22)])
        cur = None; continue
    if cur:
        m = re.match(r'^\s+([sv]_[a-z0-9_]+|ds_[a-z0-9_]+|global_[a-z0-9_]+|buffer_[a-z0-9_]+|flat_[a-z0-9_]+|scratch_[a-z0-9_]+)\b', line)
        if m: ops[m.group(1)] += 1

"""Gaps between consecutive kernels of the search loop in a rocprofv3 --kernel-trace database (graph replays)."""
import sqlite3, sys
from collections import defaultdict
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
name = 'name' if 'name' in cols else 'kernel_name'
rows = c.execute(f"select {name}, start, end from kernels order by start").fetchall()
short = lambda n: 'expand' if 'sim_expand' in n else 'mlp' if 'mlp_kernel' in n else 'finish' if 'sim_finish' in n else 'other'
gaps = defaultdict(list)
for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
    a, b = short(n0), short(n1)
    if a != 'other' and b != 'other':
        gaps[(a, b)].append(s1 - e0)
for k, v in sorted(gaps.items()):
    v.sort()
    print(f'{k[0]:>7} -> {k[1]:<7} n={len(v):5d}  median gap {v[len(v)//2]/1e3:6.2f} us  mean {sum(v)/len(v)/1e3:6.2f} us')

"""Per-move (not per-simulation) kernel time in a rocprofv3 --kernel-trace database of bench.py."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1]); moves = float(sys.argv[2])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
name = 'name' if 'name' in cols else 'kernel_name'
rows = c.execute(f"select {name}, count(*), sum(end-start) from kernels group by {name} order by 3 desc").fetchall()
tot = 0
for n, k, t in rows:
    if any(s in n for s in ('sim_expand', 'mlp_kernel', 'sim_finish')):
        continue
    tot += t
    if t / moves > 3000:
        print(f'{t/moves/1e3:8.1f} us/move  {k/moves:6.1f} calls/move  {n[:110]}')
print(f'total non-simulation kernel time per move: {tot/moves/1e3:.1f} us')

"""Per-move (not per-simulation) kernel time in a rocprofv3 --kernel-trace database of `bench.py --timed-only`: the trace is
cut into moves at bl::sim_init_kernel / bl::sim_init_env_kernel (the first launch of every search); the last `moves` complete moves -- graph replays --
are averaged.  Usage: python tools/per_move_kernels.py <db> <moves> [--sequence]   (--sequence: also the last move's launches outside the
simulations, in order)"""
import sqlite3, sys
from collections import defaultdict
c = sqlite3.connect(sys.argv[1]); want = int(float(sys.argv[2]))
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
name = 'name' if 'name' in cols else 'kernel_name'
rows = c.execute(f"select {name}, start, end from kernels order by start").fetchall()
cuts = [i for i, r in enumerate(rows) if 'sim_init_kernel' in r[0] or 'sim_init_env_kernel' in r[0]]
spans = list(zip(cuts[:-1], cuts[1:]))[-want:]
tot, calls, wall, sims = defaultdict(float), defaultdict(float), 0.0, 0.0
for a, b in spans:
    wall += rows[b][1] - rows[a][1]
    for n, s, e in rows[a:b]:
        tot[n] += e - s; calls[n] += 1
n_moves = len(spans)
glue = 0.0
print(f'{n_moves} moves, {wall / n_moves / 1e3:.1f} us from one search\'s first launch to the next')
for n, t in sorted(tot.items(), key=lambda kv: -kv[1]):
    sim = any(s in n for s in ('sim_expand', 'blmlp::mlp_kernel', 'sim_finish'))
    if not sim:
        glue += t
    if t / n_moves > 2500 or sim:
        print(f'{t / n_moves / 1e3:8.1f} us/move  {calls[n] / n_moves:6.1f} calls/move  {"[simulation] " if sim else ""}{n[:100]}')
print(f'kernel time per move outside the simulations: {glue / n_moves / 1e3:.1f} us in {sum(v for k, v in calls.items() if not any(s in k for s in ("sim_expand", "blmlp::mlp_kernel", "sim_finish"))) / n_moves:.0f} launches')
if '--sequence' in sys.argv:
    a, b = spans[-1]
    print('last move, launches outside the simulations in order (start relative to the move, duration):')
    for n, s_, e in rows[a:b]:
        if not any(x in n for x in ('sim_expand', 'blmlp::mlp_kernel', 'sim_finish')):
            print(f'  +{(s_ - rows[a][1]) / 1e3:9.1f} us {(e - s_) / 1e3:7.1f} us  {n[:150]}')

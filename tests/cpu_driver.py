"""CPU baseline driver (bench.py's cpu_baseline leg and nothing else): BASELINE config 2's search on the host cores of the
box the bench runs on, with the C oracle's kernels (kind "port": oracle/liboracle*.so, bit-checked against the reference's
CPU path by tests/test_oracle.py; nothing compiled from the reference's sources is used here).  SURVEY 8d:

  (i)  as the reference runs it: one process, single-threaded native loops over the envs;
  (ii) env-sharded over P processes = the physical cores, B/P envs each, sims/s summed (P and nproc stated);
  and ns/descent (the reference's own unit, boardlaw/mcts/tests.py:163-182) for the -O2 build and for the -O0 build
  (the reference's JIT loader passes no -O flag, boardlaw/cuda.py:29-45).

TEST/BENCH INFRASTRUCTURE -- never imported by boardlaw_amd.  Workers are tests/cpu_worker.py (numpy + ctypes, no torch)."""
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def cgroup_cpu_limit():
    """CPUs' worth of time this container may use (cgroup v2 cpu.max / v1 cfs quota), or None if unlimited."""
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        return None if quota == 'max' else float(quota) / float(period)
    except Exception:
        pass
    try:
        quota = float(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
        period = float(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
        return None if quota <= 0 else quota / period
    except Exception:
        return None


def physical_cores():
    """Cores this process can really run on at once: physical cores, capped by the affinity mask and the cgroup CPU quota
    (a GPU-pool container sees the host's 256 hardware threads in nproc but is throttled to its quota)."""
    n = len(os.sched_getaffinity(0))
    try:
        import psutil
        phys = psutil.cpu_count(logical=False)
        if phys:
            n = min(int(phys), n)
    except Exception:
        pass
    limit = cgroup_cpu_limit()
    if limit:
        n = max(1, min(n, int(limit)))
    return n


def export_weights(path, boardsize, width, depth):
    """FCModel default init under manual_seed(0), as bench.py builds it, in the worker's numpy layout."""
    import torch
    from boardlaw_amd import networks, heads
    torch.manual_seed(0)
    net = networks.FCModel(heads.Tensor((boardsize, boardsize, 2)), heads.Masked(boardsize ** 2), width=width, depth=depth)
    blocks = list(net.body)
    n = lambda t: t.detach().numpy().astype(np.float32)
    np.savez(path, w0=n(blocks[0].weight), b0=n(blocks[0].bias), wb=np.stack([n(b.weight) for b in blocks[1:]]),
             bb=np.stack([n(b.bias) for b in blocks[1:]]), alpha=np.array([float(getattr(b, 'α').detach()) for b in blocks[1:]], np.float32),
             wp=n(net.policy.core.weight), bp=n(net.policy.core.bias), wv=n(net.value.core.weight), bv=n(net.value.core.bias))


def launch(n, envs, seconds, weights, variant, boardsize, nodes, start_at):
    env = {**os.environ, 'OMP_NUM_THREADS': '1', 'OPENBLAS_NUM_THREADS': '1', 'MKL_NUM_THREADS': '1'}
    cmd = [sys.executable, os.path.join(HERE, 'cpu_worker.py'), '--boardsize', str(boardsize), '--nodes', str(nodes), '--envs', str(envs),
           '--seconds', str(seconds), '--weights', weights, '--variant', variant, '--start-at', str(start_at)]
    return [subprocess.Popen(cmd + ['--seed', str(i)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, text=True) for i in range(n)]


def collect(procs, timeout):
    out = []
    for p in procs:
        try:
            so, se = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            p.kill(); so, se = p.communicate()
        lines = [l for l in so.splitlines() if l.startswith('{')]
        if p.returncode == 0 and lines:
            out.append(json.loads(lines[-1]))
    return out


def run_cpu_baseline(boardsize, nodes, width, depth, total_envs=4096, seconds_budget=24.0, single_envs=256):
    import oracle_lib
    oracle_lib.load(''); oracle_lib.load('_O0')            # build the checker libraries before the workers race for them
    P, nproc = physical_cores(), os.cpu_count()
    with tempfile.TemporaryDirectory() as tmp:
        weights = os.path.join(tmp, 'fcmodel.npz')
        export_weights(weights, boardsize, width, depth)
        t_par, t_one = 0.45 * seconds_budget, 0.15 * seconds_budget
        envs = max(1, total_envs // P)
        # (ii) all physical cores, started together
        procs = launch(P, envs, t_par, weights, '', boardsize, nodes, time.time() + 3.0)
        par = collect(procs, timeout=t_par * 4 + 60)
        # (i) one process, and the -O0 build, side by side on two otherwise idle cores
        procs = launch(1, single_envs, t_one, weights, '', boardsize, nodes, time.time() + 1.5) + \
            launch(1, single_envs, t_one, weights, '_O0', boardsize, nodes, time.time() + 1.5)
        one = collect(procs, timeout=t_one * 6 + 60)
    if not par or len(one) < 2:
        return {'value': None, 'unit': 'sims/s', 'cores': P, 'nproc': nproc, 'kind': 'port', 'sample': 'workers failed'}
    rate = sum(r['sims'] / r['seconds'] for r in par)
    ns = lambda r: 1e9 * r['descend_seconds'] / max(r['descents'], 1)
    o2, o0 = one
    return {
        'value': rate, 'unit': 'sims/s', 'cores': len(par), 'nproc': nproc, 'kind': 'port',
        'cgroup_cpu_limit': cgroup_cpu_limit(),
        'sample': f'{len(par)} processes (one per usable physical core: min(physical cores, affinity, cgroup CPU quota); nproc {nproc}) x {envs} envs x {nodes} sims/move, {boardsize}x{boardsize}, '
                  f'FCModel {width}x{depth} f32 (numpy, 1 BLAS thread each), C oracle -O2, {t_par:.0f} s each after a warm-up move '
                  f'({sum(r["moves"] for r in par)} moves in total)',
        'single_thread': {'value': o2['sims'] / o2['seconds'], 'unit': 'sims/s', 'envs': o2['envs'], 'ns_per_descent_O2': ns(o2),
                          'ns_per_descent_O0': ns(o0), 'value_O0': o0['sims'] / o0['seconds'],
                          'note': f'one process, {single_envs} envs; -O0 = how the reference JIT-builds its sources (no -O flag, libm powf)'},
        'ns_per_descent_parallel_O2': float(np.mean([ns(r) for r in par])),
    }

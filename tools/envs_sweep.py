"""sims/s against the number of parallel envs -- the reference's `benchmark_mcts` (boardlaw/mcts/tests.py:323-348: runtime of a
search-agent rollout for n_envs = 2^0 .. 2^14) and its descent benchmark (mcts/tests.py:163-182: ns/descent of `cuda.descend`)
restated for the path's own shape: 9x9 Hex, 64 sims per move, FCModel 512x4, n_envs = 2^8 .. 2^15 on one MI355X.  2^15 = 32 768
envs is the reference's operating point (boardlaw/main.py:147: `n_envs=32*1024`); BASELINE config 2 quotes the metric at 2^12.

    python tools/envs_sweep.py [--lo 8] [--hi 15] [--steps 10] [--boardsize 9] [--nodes 64] [--width 512] [--depth 4]

Per point: sims/s of captured self-play moves (bench.py's `value` at that batch), the mean duration of `bl_sim_expand` and of
`bl_sim_infer_finish` under HIP events in an eager re-run of the same moves, d / k (policy evaluations and child lookups per
descent, counted by the kernel), bl_sim_expand's algorithmic GB/s against the 8 TB/s peak (bench.py's `roofline`), and GPU
ns/descent = bl_sim_expand's duration / envs (one descent per env per launch; the kernel also expands, steps and observes the
leaf, so this is an upper bound on the descent alone) -- the unit the reference prints, beside bench.py's
`cpu_baseline.ns_per_descent_*`."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def point(B, args):
    from boardlaw_amd import _native, networks
    from boardlaw_amd.hex import Hex
    from boardlaw_amd.mcts import MCTSAgent, MoveRng
    lib = _native.lib()
    S, T = args.boardsize, args.nodes
    bench.BOARD, bench.NODES = S, T
    gen = torch.Generator(device='cuda'); gen.manual_seed(1000)
    torch.manual_seed(0)
    worlds = Hex.initial(B, S)
    net = networks.FCModel(worlds.obs_space, worlds.action_space, width=args.width, depth=args.depth).cuda()
    worlds = bench.premix(worlds, S * S // 3, gen)
    torch.manual_seed(1)
    inf = networks.Inference(net, fused=True)
    agent = MCTSAgent(inf, n_nodes=T, graph=True, rng=MoveRng())
    sync = torch.cuda.synchronize
    rate = bench.rate_of(agent, worlds, args.steps, 2, sync)
    del agent
    # the same moves launched kernel by kernel, both search kernels under HIP events
    fused = inf.prefers_fused(B)
    t_exp, t_fin = bench.TimedExpand(lib), bench.TimedExpand(lib, 'bl_sim_infer_finish' if fused else 'bl_sim_finish')
    lib.bl_sim_expand = t_exp
    setattr(lib, 'bl_sim_infer_finish' if fused else 'bl_sim_finish', t_fin)
    try:
        probe = MCTSAgent(inf, n_nodes=T, graph=False, rng=MoveRng())
        w = probe.play(worlds)[1]
        t_exp.on = t_fin.on = True
        for _ in range(args.probe_moves):
            w = probe.play(w)[1]
        sync()
    finally:
        lib.bl_sim_expand = t_exp.orig
        setattr(lib, 'bl_sim_infer_finish' if fused else 'bl_sim_finish', t_fin.orig)
    d, k, its = bench.tree_statistics(worlds, net, T)
    A = S * S
    exp_us, fin_us = t_exp.mean_us(), t_fin.mean_us()
    per_launch = bench.expand_bytes_per_env(A, 2, d, k) * B
    achieved = per_launch / (exp_us * 1e-6) / 1e9
    whole = bench.total_bytes_per_sim(A, 2, T, d, k)
    res = {'n_envs': B, 'sims_per_sec': rate, 'ms_per_move': 1e3 * B * T / rate, 'bl_sim_expand_us': exp_us,
           ('bl_sim_infer_finish_us' if fused else 'bl_sim_finish_us'): fin_us, 'network_plan': 'fused' if fused else 'layers',
           'gpu_ns_per_descent': 1e3 * exp_us / B, 'd': round(d, 3), 'k': round(k, 3), 'newton_iters': round(its, 3),
           'expand_GBs': achieved, 'expand_frac_of_hbm': achieved / bench.HBM_PEAK_GBS,
           'whole_path_frac_of_hbm': whole * rate / (bench.HBM_PEAK_GBS * 1e9)}
    del probe, w, worlds, net, inf
    torch.cuda.empty_cache()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--lo', type=int, default=8)
    ap.add_argument('--hi', type=int, default=15)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--probe-moves', type=int, default=2)
    ap.add_argument('--boardsize', type=int, default=9)
    ap.add_argument('--nodes', type=int, default=64)
    ap.add_argument('--width', type=int, default=512)
    ap.add_argument('--depth', type=int, default=4)
    args = ap.parse_args()
    assert torch.cuda.is_available(), 'envs_sweep.py needs an MI355X'
    print(f'# {args.boardsize}x{args.boardsize} Hex, {args.nodes} sims/move, FCModel {args.width}x{args.depth}, one MI355X; harness: boardlaw/mcts/tests.py:323-348 '
          f'(runtime vs n_envs) and :163-182 (ns/descent)')
    print(f'{"n_envs":>7} {"M sims/s":>9} {"ms/move":>8} {"expand us":>10} {"finish us":>10} {"ns/descent":>11} {"d":>6} {"k":>6} {"expand GB/s":>12} {"of 8 TB/s":>10} {"path of HBM":>12}')
    rows = []
    for e in range(args.lo, args.hi + 1):
        r = point(1 << e, args)
        rows.append(r)
        fin = r.get('bl_sim_infer_finish_us', r.get('bl_sim_finish_us'))
        print(f'{r["n_envs"]:>7} {r["sims_per_sec"] / 1e6:>9.2f} {r["ms_per_move"]:>8.2f} {r["bl_sim_expand_us"]:>10.1f} {fin:>10.1f} {r["gpu_ns_per_descent"]:>11.2f} '
              f'{r["d"]:>6.2f} {r["k"]:>6.2f} {r["expand_GBs"]:>12.1f} {100 * r["expand_frac_of_hbm"]:>9.2f}% {100 * r["whole_path_frac_of_hbm"]:>11.2f}%', flush=True)
    print(json.dumps({'envs_sweep': rows}))


if __name__ == '__main__':
    main()

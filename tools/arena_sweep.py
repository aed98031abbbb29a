"""BASELINE config 5 -- the mixed board-size arena sweep (boards 3..11, 2048 games each between two 64-sim search agents,
arena.evaluate's masked calls) -- fanned out like the reference fans out its evaluation (boardlaw/arena/neural.py:257-274 over
rebar/parallel.py:28-57): one job per board size (x --repeat), a pool of worker processes, worker n on GPU n % n_gpus.
`--workers` defaults to arena.workers_per_gpu(envs) per GPU -- three for matches of up to 2048 games: independent matches side by
side on a GPU (the search kernels are latency-bound, so further resident matches fill cycles the first leaves idle -- DESIGN.md 5).

    python tools/arena_sweep.py [--gpus N] [--workers W] [--boards 3,5,7,9,11] [--envs 2048] [--repeat 1] [--eager]

Prints one line per finished match and a final JSON line with the whole sweep's games/s and sims/s.
ARENA_DRY=1: the same fan-out with deterministic agents on CPU worlds (tests/pool_helpers.py), no GPU -- the N > 1 path's test."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


def search_match(S, n_envs, nodes, width, depth, graph, start_at=None):
    """One arena match between two search agents on this worker's device; the second of two back-to-back matches is timed
    (the first captures the moves / warms the library up).  start_at (wall clock): every worker's timed match starts then, so
    that concurrent matches really overlap."""
    import torch
    from boardlaw_amd import arena, networks
    from boardlaw_amd.hex import Hex
    from boardlaw_amd.mcts import MCTSAgent, MoveRng
    pair = {}
    for i, name in enumerate(('one', 'two')):
        torch.manual_seed(i)
        w0 = Hex.initial(1, S)
        pair[name] = MCTSAgent(networks.Inference(networks.FCModel(w0.obs_space, w0.action_space, width, depth).cuda(), fused=True),
                               graph=graph, n_nodes=nodes, rng=MoveRng())
    for rep in range(2):
        if rep == 1 and start_at is not None and time.time() < start_at:
            time.sleep(start_at - time.time())
        torch.cuda.synchronize(); t0 = time.time()
        res = arena.evaluate(Hex.initial(n_envs, S), pair)
        torch.cuda.synchronize(); dt = time.time() - t0
    return dict(S=S, device=torch.cuda.current_device(), seconds=dt, games=sum(r.games for r in res), moves=sum(r.moves for r in res),
                wins=[list(r.wins) for r in res], start=t0, end=t0 + dt)


def dry_match(S, n_envs, *_):
    import pool_helpers
    from boardlaw_amd import arena
    t0 = time.time()
    res = arena.evaluate(pool_helpers.cpu_worlds(n_envs, S), {'one': pool_helpers.edge_agent('e0'), 'two': pool_helpers.edge_agent('e3')})
    return dict(S=S, device=-1, seconds=time.time() - t0, games=sum(r.games for r in res), moves=sum(r.moves for r in res),
                wins=[list(r.wins) for r in res], start=t0, end=time.time())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=None, help='GPUs to use (default: all visible)')
    ap.add_argument('--workers', type=int, default=None, help='worker processes (default: arena.workers_per_gpu(envs) per GPU: three for matches of up to 2048 games)')
    ap.add_argument('--boards', type=str, default='3,5,7,9,11')
    ap.add_argument('--envs', type=int, default=2048)
    ap.add_argument('--nodes', type=int, default=64)
    ap.add_argument('--width', type=int, default=512)
    ap.add_argument('--depth', type=int, default=4)
    ap.add_argument('--repeat', type=int, default=1, help='matches per board size')
    ap.add_argument('--eager', action='store_true', help='no captured moves')
    ap.add_argument('--together', type=float, default=0., help='seconds from now at which every worker starts its timed match (use with one '
                                                              'job per worker to measure matches that overlap completely); 0: as they come')
    args = ap.parse_args()
    dry = os.environ.get('ARENA_DRY') == '1'
    import torch
    from boardlaw_amd import arena
    if args.gpus is not None and not dry:
        os.environ['HIP_VISIBLE_DEVICES'] = ','.join(str(i) for i in range(args.gpus))
    n_gpus = 0 if dry else torch.cuda.device_count()
    workers = args.workers if args.workers is not None else max(n_gpus, 1) * arena.workers_per_gpu(args.envs)
    boards = [int(x) for x in args.boards.split(',')]
    fn = dry_match if dry else search_match
    start_at = time.time() + args.together if args.together > 0 else None
    jobs = {(S, r): (fn, (S, args.envs, args.nodes, args.width, args.depth, not args.eager, start_at)) for r in range(args.repeat) for S in sorted(boards, reverse=True)}
    os.environ['PYTHONPATH'] = os.pathsep.join([ROOT, os.path.join(ROOT, 'tests'), os.path.dirname(os.path.abspath(__file__)), os.environ.get('PYTHONPATH', '')])
    t0 = time.time()
    done = []
    for key, r in arena.run_jobs(jobs, n_workers=workers):
        done.append(r)
        print(f'board {r["S"]}x{r["S"]} on device {r["device"]}: {r["games"]:.0f} games, {r["moves"]:.0f} moves in {r["seconds"]:.2f} s '
              f'({r["moves"] * args.nodes / r["seconds"] / 1e6:.2f} M sims/s alone)', flush=True)
    wall = time.time() - t0
    # the window in which matches were actually being played (process start-up and capture excluded): first timed start to last end
    span = max(r['end'] for r in done) - min(r['start'] for r in done)
    moves, games = sum(r['moves'] for r in done), sum(r['games'] for r in done)
    print(json.dumps({'sweep': 'arena', 'boards': boards, 'envs': args.envs, 'nodes': args.nodes, 'network': f'{args.width}x{args.depth}', 'gpus': n_gpus, 'workers': workers,
                      'matches': len(done), 'games': games, 'moves': moves, 'wall_s': wall, 'timed_span_s': span, 'games_per_sec': games / span,
                      'sims_per_sec': moves * args.nodes / span, 'captured_moves': not args.eager, 'dry_run': dry,
                      'devices_used': sorted({r['device'] for r in done})}))


if __name__ == '__main__':
    main()

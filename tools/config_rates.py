"""BASELINE.json's other configurations on ONE GPU (the bench line is config 2): config 1 (5x5, 64 envs, 16 sims), config 4's
per-GPU shape (13x13, 1024 envs, 256 sims, 1024x8) and config 5's arena sweep (boards 3..11, 2048 envs each, two 64-sim agents,
arena.evaluate's masked calls).  Self-play rows: sims/s over `--moves` captured moves after 3 warm-up moves; arena rows: one
match of 2048 games.  usage (GPU box): python tools/config_rates.py [--moves 30]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from boardlaw_amd import arena, networks
from boardlaw_amd.hex import Hex
from boardlaw_amd.mcts import MCTSAgent, MoveRng
from bench import premix

# (10 moves of config 2 are 58 ms: too short a window on a box whose clocks have just idled through a capture)
ap = argparse.ArgumentParser(); ap.add_argument('--moves', type=int, default=30); args = ap.parse_args()
gen = torch.Generator(device='cuda'); gen.manual_seed(0)


def selfplay(S, B, T, width, depth, label):
    torch.manual_seed(0)
    worlds = premix(Hex.initial(B, S), (S * S) // 3, gen)
    net = networks.Inference(networks.FCModel(worlds.obs_space, worlds.action_space, width, depth).cuda(), fused=True)
    agent = MCTSAgent(net, n_nodes=T, graph=True, rng=MoveRng())
    for _ in range(3):
        agent(worlds)
    # at least --moves moves AND half a second: config 1's 30 moves are 27 ms, a window in which a box whose clocks idled through
    # the capture reads 2x slow (this file's config-1 and config-2 rows have both shown that)
    torch.cuda.synchronize(); t0 = time.time(); n = 0
    while n < args.moves or time.time() - t0 < 0.5:
        for _ in range(args.moves):
            agent(worlds)
        torch.cuda.synchronize(); n += args.moves
    dt = (time.time() - t0) / n
    if label is None:
        return
    plan = 'library GEMMs' if net._packed is None else 'one kernel' if net.prefers_fused(B) else 'launch per Linear'
    print(f'{label}: {S}x{S}, {B} envs x {T} sims, FCModel {width}x{depth} ({plan}): {1e3 * dt:.2f} ms per move, {B * T / dt / 1e6:.2f} M sims/s', flush=True)


selfplay(9, 4096, 64, 512, 4, None)          # unmeasured: the process's first captures and the GPU's clocks
selfplay(5, 64, 16, 16, 4, 'config 1')
selfplay(9, 4096, 64, 512, 4, 'config 2 (bench.py is the measurement)')
selfplay(13, 1024, 256, 1024, 8, 'config 4, per GPU')
for S in (3, 5, 7, 9, 11):
    for mode in ('eager', 'captured'):
        pair = {}
        for name in ('one', 'two'):
            torch.manual_seed(len(pair))
            w0 = Hex.initial(1, S)
            net = networks.Inference(networks.FCModel(w0.obs_space, w0.action_space, 512, 4).cuda(), fused=True)
            pair[name] = MCTSAgent(net, graph=(mode == 'captured'), n_nodes=64, rng=MoveRng())
        for rep in range(2):          # captured: the first match captures one move per capacity bucket, the second replays them
            worlds = Hex.initial(2048, S)
            torch.cuda.synchronize(); t0 = time.time()
            res = arena.evaluate(worlds, pair)
            torch.cuda.synchronize(); dt = time.time() - t0
        moves = sum(r.moves for r in res)
        print(f'config 5, per GPU: arena {S}x{S}, 2048 games, two 64-sim agents (512x4), {mode} moves: {dt:.2f} s, {sum(r.games for r in res) / dt:.0f} games/s, '
              f'{moves * 64 / dt / 1e6:.2f} M sims/s over {moves:.0f} moves', flush=True)

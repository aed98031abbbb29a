"""One worker process of bench.py's cpu_baseline leg: the search of BASELINE config 2 on the host, single-threaded, with
the C oracle's kernels (oracle/liboracle*.so) and the network in numpy -- no torch in here, so that one worker per
physical core fits in memory.  TEST/BENCH INFRASTRUCTURE: never imported by boardlaw_amd.  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle_lib  # noqa: E402
from oracle_lib import OracleSearch, f16_bits  # noqa: E402


class NumpyFCModel:
    """boardlaw/networks.py:10-40 + heads.py:47-52,101-104,128-142 in f32 numpy (the weights come from the torch module)."""

    def __init__(self, z):
        self.w0, self.b0 = np.ascontiguousarray(z['w0'].T), z['b0']
        self.wb = [np.ascontiguousarray(w.T) for w in z['wb']]
        self.bb, self.alpha = z['bb'], z['alpha']
        self.wp, self.bp = np.ascontiguousarray(z['wp'].T), z['bp']
        self.wv, self.bv = np.ascontiguousarray(z['wv'].T), z['bv']

    def __call__(self, obs, valid, seats):
        x = obs.reshape(obs.shape[0], -1) @ self.w0 + self.b0
        for w, b, a in zip(self.wb, self.bb, self.alpha):
            x = x + a * (np.maximum(x, 0) @ w + b)
        raw = np.where(valid, x @ self.wp + self.bp, -np.inf).astype(np.float32)
        mx = raw.max(-1, keepdims=True)
        logits = raw - mx - np.log(np.exp(raw - mx).sum(-1, keepdims=True))
        v = np.tanh(x @ self.wv + self.bv)[:, 0]
        vv = np.stack([v, -v], -1)
        vv = np.where(seats[:, None] == 0, vv, vv[:, ::-1])
        return logits.astype(np.float32), vv.astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--boardsize', type=int, default=9)
    ap.add_argument('--nodes', type=int, default=64)
    ap.add_argument('--envs', type=int, default=64)
    ap.add_argument('--seconds', type=float, default=8.)
    ap.add_argument('--weights', required=True)
    ap.add_argument('--variant', default='', help="'' = liboracle.so (-O2), '_O0' = liboracle_O0.so")
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--start-at', type=float, default=0., help='unix time at which all workers start their clocks')
    args = ap.parse_args()
    kern = oracle_lib.load(args.variant)
    net = NumpyFCModel(np.load(args.weights, allow_pickle=True))
    S, A, envs, nodes = args.boardsize, args.boardsize ** 2, args.envs, args.nodes
    rng = np.random.default_rng(args.seed)
    board = np.zeros((envs, S, S), np.uint8); seats = np.zeros(envs, np.int32)
    for _ in range(A // 3):                               # bench.py's premix: floor(S^2/3) random legal moves
        valid = (kern.hex_observe(board, seats) == 0).all(-1).reshape(envs, -1)
        board, seats, _, _ = kern.hex_world_step(board, seats, (rng.random(valid.shape) * valid).argmax(-1).astype(np.int32))

    def evaluate(b, s):
        obs = kern.hex_observe(b, s)
        return net(obs, (obs == 0).all(-1).reshape(len(b), -1), s)

    t_descend, n_descents = 0., 0

    def one_move(board, seats):
        nonlocal t_descend, n_descents
        s = OracleSearch(kern, board, seats, nodes)
        logits, v = evaluate(board, seats)
        s.initialize(f16_bits(logits), f16_bits(v))
        for _ in range(nodes - 1):
            rands = f16_bits(rng.random((envs, nodes), dtype=np.float32))
            t0 = time.perf_counter()
            parents, actions = s.descend(rands)
            t_descend += time.perf_counter() - t0; n_descents += envs
            leaves, nb, ns = s.expand(parents, actions)
            logits, v = evaluate(nb, ns)
            s.finish(leaves, f16_bits(logits), f16_bits(v))
        probs = oracle_lib.f16_vals(s.root_probs())
        actions = (probs + rng.random(probs.shape) * 1e-3 * (probs > 0)).argmax(-1).astype(np.int32)
        nb, ns, _, _ = kern.hex_world_step(board, seats, actions)
        return nb, ns

    board, seats = one_move(board, seats)                 # warm-up move (page in, BLAS init)
    t_descend, n_descents = 0., 0
    while time.time() < args.start_at:
        time.sleep(0.01)
    moves, t0 = 0, time.perf_counter()
    while True:
        board, seats = one_move(board, seats)
        moves += 1
        el = time.perf_counter() - t0
        if el >= args.seconds:
            break
    print(json.dumps({'sims': envs * nodes * moves, 'seconds': el, 'moves': moves, 'envs': envs,
                      'descend_seconds': t_descend, 'descents': n_descents}))


if __name__ == '__main__':
    main()

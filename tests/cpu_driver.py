"""CPU baseline driver (bench.py's cpu_baseline leg and nothing else): the same search loop as the product, run on the
host with the reference's own CPU kernels (oracle/_ref/*.so, the reference's unmodified sources compiled by
oracle/Makefile) when present, else with the C oracle.  TEST/BENCH INFRASTRUCTURE -- never imported by boardlaw_amd."""
import importlib.util
import os
import time

import numpy as np
import torch

import oracle_lib
from oracle_lib import OracleSearch, f16_bits

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _h(bits):
    return torch.from_numpy(bits.view(np.int16)).view(torch.half)


class RefKernels:
    """oracle_lib.Oracle's numpy interface on top of the compiled reference modules (zero-copy tensor views)."""

    def __init__(self):
        def load(name):
            spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, 'oracle', '_ref', name + '.so'))
            m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
            return m
        self.m, self.h = load('mctscuda'), load('hexcuda')

    def _tree(self, logits, w, n, c_puct, seats, terminal, children):
        return self.m.MCTS(_h(logits), _h(w), torch.from_numpy(n), _h(c_puct), torch.from_numpy(seats),
                           torch.from_numpy(terminal).view(torch.bool), torch.from_numpy(children))

    def descend(self, logits, w, n, c_puct, seats, terminal, children, rands=None, stats=None):
        d = self.m.descend(self._tree(logits, w, n, c_puct, seats, terminal, children))   # draws its own rands (cpu.cpp:187)
        return d.parents.numpy(), d.actions.numpy()

    def root(self, *tree):
        return self.m.root(self._tree(*tree)).view(torch.int16).numpy().view(np.uint16)

    def backup(self, v, w, n, rewards, parents, terminal, leaves):
        bk = self.m.Backup(v=_h(v), w=_h(w), n=torch.from_numpy(n), rewards=_h(rewards), parents=torch.from_numpy(parents),
                           terminal=torch.from_numpy(terminal).view(torch.bool))
        self.m.backup(bk, torch.from_numpy(leaves))

    def hex_observe(self, board, seats):
        return self.h.observe(torch.from_numpy(board), torch.from_numpy(seats)).numpy()

    def hex_world_step(self, board, seats, actions):
        nb = torch.from_numpy(board.copy())
        rewards = self.h.step(nb, torch.from_numpy(seats), torch.from_numpy(actions)).numpy()
        term = (rewards > 0).any(-1)
        nb = nb.numpy(); nb[term] = 0
        ns = np.where(term, 0, 1 - seats).astype(np.int32)
        return nb, ns, rewards, term.astype(np.uint8)


class _World:
    pass


def run_cpu_search(boardsize, nodes, width, depth, seconds_budget=25.0, envs=256):
    from boardlaw_amd import networks, heads
    threads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        try:
            kern, kind = RefKernels(), 'reference'
        except Exception:
            kern, kind = oracle_lib.load(), 'port'
        S, A = boardsize, boardsize * boardsize
        torch.manual_seed(0)
        net = networks.FCModel(heads.Tensor((S, S, 2)), heads.Masked(A), width=width, depth=depth)
        rng = np.random.default_rng(0)
        board = np.zeros((envs, S, S), np.uint8); seats = np.zeros(envs, np.int32)
        for _ in range(A // 3):
            valid = (kern.hex_observe(board, seats) == 0).all(-1).reshape(envs, -1)
            board, seats, _, _ = kern.hex_world_step(board, seats, (rng.random(valid.shape) * valid).argmax(-1).astype(np.int32))

        def evaluate(b, s):
            w = _World()
            w.obs = torch.from_numpy(kern.hex_observe(b, s)); w.valid = (w.obs == 0).all(-1).reshape(len(b), -1)
            w.seats = torch.from_numpy(s)
            with torch.no_grad():
                d = net(w)
            return d.logits, d.v

        def one_move(board, seats):
            s = OracleSearch(kern, board, seats, nodes)
            logits, v = evaluate(board, seats)
            s.initialize(f16_bits(logits.numpy()), f16_bits(v.numpy()))
            for _ in range(nodes - 1):
                rands = f16_bits(rng.random((envs, nodes), dtype=np.float32))
                parents, actions = s.descend(rands)
                leaves, nb, ns = s.expand(parents, actions)
                logits, v = evaluate(nb, ns)
                s.finish(leaves, f16_bits(logits.numpy()), f16_bits(v.numpy()))
            probs = oracle_lib.f16_vals(s.root_probs())
            actions = (probs + rng.random(probs.shape) * 1e-3 * (probs > 0)).argmax(-1).astype(np.int32)
            nb, ns, _, _ = kern.hex_world_step(board, seats, actions)
            return nb, ns

        moves, t0 = 0, time.perf_counter()
        while True:
            board, seats = one_move(board, seats)
            moves += 1
            el = time.perf_counter() - t0
            if el > 0.6 * seconds_budget or el * (moves + 1) / moves > seconds_budget:
                break
        return {'value': envs * nodes * moves / el, 'unit': 'sims/s', 'cores': 1, 'kind': kind,
                'sample': f'{moves} move(s) of {envs} envs x {nodes} sims, {boardsize}x{boardsize}, FCModel {width}x{depth} fp32, '
                          f'1 thread, {"reference CPU sources built -O2 (oracle/_ref)" if kind == "reference" else "C oracle -O2"}; '
                          f'{el:.1f}s'}
    finally:
        torch.set_num_threads(threads)

"""Seat-permuted evaluation matches between agents (boardlaw/arena/common.py:52-106): every env is assigned one
permutation of the agents over the seats; each round every agent moves in the envs where it is its turn
(`agent(worlds[mask], eval=True)` -- variable batch size, argmax actions), until every env has finished one game.
Returns per permutation: names, wins per seat, moves, games, wall time."""
import math
import time
from itertools import permutations

import numpy as np
import torch

from . import arrdict


def matchup_patterns(n_seats):
    return torch.as_tensor(list(permutations(range(n_seats))))


def matchup_indices(n_envs, n_seats):
    patterns = matchup_patterns(n_seats)
    return patterns.repeat((n_envs // len(patterns), 1))


def _tally(seat_wins, n_moves, seconds, assignment, names, boardsize):
    """One record per seat permutation: who sat where, wins per seat, moves, games, wall time."""
    names = np.array(names)
    records = []
    for perm in matchup_patterns(assignment.shape[1]):
        rows = (assignment == perm).all(-1)
        w = seat_wins[rows].sum(0)
        records.append(arrdict.dotdict(names=tuple(names[perm]), wins=tuple(float(x) for x in w), moves=float(n_moves[rows].sum()),
                                       games=float(w.sum()), times=float(seconds[rows].sum()), boardsize=boardsize))
    return records


def evaluate(worlds, agents):
    """Every env plays ONE game; env e seats the agents in the order `assignment[e]` (all seat permutations, tiled over
    the batch).  A round lets each agent act, with argmax actions, in exactly the unfinished envs where it is to move."""
    roster = list(agents.items()) if isinstance(agents, dict) else list(agents)
    n_seats, n_envs, dev = worlds.n_seats, worlds.n_envs, worlds.device
    assert n_seats == 2, 'Only support 2 seats for now'
    assert n_envs % math.factorial(n_seats) == 0, 'Number of envs needs to be divisible by the number of permutations of seats'
    assert len(roster) == n_seats, 'Need to pass one agent per seat'
    assignment = matchup_indices(n_envs, n_seats).to(dev)
    everyone = torch.arange(n_envs, device=dev)
    done = torch.zeros(n_envs, dtype=torch.bool, device=dev)
    seat_wins = torch.zeros((n_envs, n_seats), dtype=torch.int, device=dev)
    n_moves = torch.zeros(n_envs, dtype=torch.int, device=dev)
    seconds = torch.zeros(n_envs, dtype=torch.float, device=dev)
    while not bool(done.all()):
        for who, (_, agent) in enumerate(roster):
            to_move = (assignment[everyone, worlds.seats.long()] == who) & ~done
            if not bool(to_move.any()):
                continue
            t0 = time.time()
            picks = agent(worlds[to_move], eval=True).actions
            worlds[to_move], outcome = worlds[to_move].step(picks)
            done[to_move] = outcome.terminal
            elapsed = time.time() - t0
            seat_wins[to_move] += (outcome.rewards == 1).int()
            n_moves[to_move] += 1
            seconds[to_move] += elapsed / to_move.sum()
    return _tally(seat_wins.cpu(), n_moves.cpu(), seconds.cpu(), assignment.cpu(), [name for name, _ in roster],
                  getattr(worlds, 'boardsize', None))


from .analysis import rollout  # noqa: E402,F401  (kept here for callers that imported it from arena)

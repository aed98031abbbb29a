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


def gather(wins, moves, times, matchup_idxs, agents, boardsize):
    names = np.array([name for name, _ in agents])
    results = []
    for p in matchup_patterns(matchup_idxs.shape[1]):
        sel = (matchup_idxs == p).all(-1)
        ws = wins[sel].sum(0)
        results.append(arrdict.dotdict(names=tuple(names[p]), wins=tuple(map(float, ws)), moves=float(moves[sel].sum(0)),
                                       games=float(ws.sum()), times=float(times[sel].sum(0)), boardsize=boardsize))
    return results


def evaluate(worlds, agents):
    if isinstance(agents, dict):
        agents = list(agents.items())
    assert worlds.n_seats == 2, 'Only support 2 seats for now'
    assert worlds.n_envs % math.factorial(worlds.n_seats) == 0, 'Number of envs needs to be divisible by the number of permutations of seats'
    assert len(agents) == worlds.n_seats, 'Need to pass one agent per seat'
    dev, B = worlds.device, worlds.n_envs
    envs = torch.arange(B, device=dev)
    terminal = torch.zeros((B,), dtype=torch.bool, device=dev)
    wins = torch.zeros((B, worlds.n_seats), dtype=torch.int, device=dev)
    moves = torch.zeros((B,), dtype=torch.int, device=dev)
    times = torch.zeros((B,), dtype=torch.float, device=dev)
    matchup_idxs = matchup_indices(B, worlds.n_seats).to(dev)
    while True:
        for i, (_, agent) in enumerate(agents):
            mask = (matchup_idxs[envs, worlds.seats.long()] == i) & ~terminal
            if mask.any():
                start = time.time()
                decisions = agent(worlds[mask], eval=True)
                worlds[mask], transitions = worlds[mask].step(decisions.actions)
                terminal[mask] = transitions.terminal
                end = time.time()
                wins[mask] += (transitions.rewards == 1).int()
                moves[mask] += 1
                times[mask] += (end - start) / mask.sum()
        if terminal.all():
            break
    return gather(wins.cpu(), moves.cpu(), times.cpu(), matchup_idxs.cpu(), agents, getattr(worlds, 'boardsize', None))


def combine_actions(decisions, masks):
    actions = torch.cat([d.actions for d in decisions.values()])
    for mask, decision in zip(masks.values(), decisions.values()):
        actions[mask] = decision.actions
    return actions


@torch.no_grad()
def rollout(worlds, agents, n_steps=None, n_trajs=None, n_reps=None, **kwargs):
    """Plays the agents against each other by seat and records the trace (boardlaw/analysis.py:47-87):
    arrdict(actions, transitions, worlds) stacked over time."""
    assert sum(x is not None for x in (n_steps, n_trajs, n_reps)) == 1, 'Must specify exactly one of n_steps or n_trajs or n_reps'
    trace = []
    steps, trajs = 0, 0
    reps = torch.zeros(worlds.n_envs, device=worlds.device)
    while True:
        decisions, masks = {}, {}
        for i, agent in enumerate(agents):
            mask = worlds.seats == i
            if mask.any():
                decisions[i] = agent(worlds[mask], **kwargs)
                masks[i] = mask
        actions = combine_actions(decisions, masks)
        worlds, transitions = worlds.step(actions)
        trace.append(arrdict.arrdict(actions=actions, transitions=transitions, worlds=worlds))
        steps += 1
        if n_steps and steps >= n_steps:
            break
        trajs += transitions.terminal.sum()
        if n_trajs and trajs >= n_trajs:
            break
        reps += transitions.terminal
        if n_reps and (reps >= n_reps).all():
            break
    return arrdict.stack(trace)

"""The actor/learner loop around the search (boardlaw/main.py:61-98,147-200), without the reference's run directory,
stats, storage and live-arena side effects: collect `buffer_len` moves of self-play, turn them into a chunk with
reward-to-go targets, take one AMP Adam step on one random timestep per env.

Multi-GPU: one process per GPU, each with its own env shard; the only collective is the gradient all-reduce
(`parallel.allreduce_gradients`: one flat bucket over RCCL -- the 512x4 network is 1.2 M parameters, a single 4.7 MB
fp32 bucket, latency- not bandwidth-bound on xGMI's point-to-point links, so bucketing finer would only add launches).
The reference is single-device; with world size 1 this is its `optimize`."""
import numpy as np
import torch

from . import arrdict, learning, parallel


def as_chunk(buffer, batch_size):
    """main.py:61-74: stack the buffer over time, attach reward_to_go, drop the oldest batch_size/B steps."""
    chunk = arrdict.stack(buffer)
    n_seats = chunk.worlds.n_seats if hasattr(chunk.worlds, 'n_seats') else chunk.decisions.v.shape[-1]
    terminal = torch.stack([chunk.transitions.terminal for _ in range(n_seats)], -1)
    chunk['reward_to_go'] = learning.reward_to_go(
        chunk.transitions.rewards.float(), chunk.decisions.v.float(), terminal).half()
    n_new = batch_size // terminal.size(1)
    return chunk, buffer[n_new:]


def losses(network, batch):
    """Policy cross-entropy against the search's root distribution + value MSE against reward-to-go (main.py:78-92)."""
    d0 = batch.decisions
    d = network(batch.worlds)
    zeros = torch.zeros_like(d.logits)
    l = d.logits.where(d.logits > -np.inf, zeros)
    l0 = d0.logits.float().where(d0.logits > -np.inf, zeros.float())
    policy_loss = -(l0.exp() * l).sum(-1).mean()
    value_loss = (batch.reward_to_go - d.v).square().mean()
    return policy_loss, value_loss


def optimize(network, scaler, opt, batch, sync_gradients=True):
    """One learner step (main.py:76-98).  Returns (policy_loss, value_loss) as detached tensors."""
    cuda = next(network.parameters()).is_cuda
    with torch.autocast('cuda', enabled=cuda):
        policy_loss, value_loss = losses(network, batch)
        loss = policy_loss + value_loss
    opt.zero_grad()
    scaler.scale(loss).backward()
    if sync_gradients:
        parallel.allreduce_gradients(network)
    scaler.step(opt)
    scaler.update()
    return policy_loss.detach(), value_loss.detach()


def run(worlds, network, n_steps, nodes=64, c_puct=1 / 16, lr=1e-3, buffer_len=64, graph=False, inference=None, on_step=None):
    """main.py:147-200 minus run bookkeeping.  `worlds` is this rank's env shard.  Returns the final worlds.
    inference: None (the module under autocast, as the reference), 'torch' or 'fused' (networks.Inference plans; their
    f16 weights are refreshed from the module at the start of every move, so optimiser steps are seen)."""
    from . import networks
    from .mcts import MCTSAgent
    n_envs, dev = worlds.n_envs, worlds.device
    actor = network if inference is None else networks.Inference(network, fused=(inference == 'fused'))
    agent = MCTSAgent(actor, n_nodes=nodes, c_puct=c_puct, graph=graph)
    opt = torch.optim.Adam(network.parameters(), lr=lr)
    scaler = torch.amp.GradScaler('cuda', enabled=(dev.type == 'cuda'))
    idxs = (torch.randint(buffer_len, (n_envs,), device=dev), torch.arange(n_envs, device=dev))
    buffer = []
    for step in range(n_steps):
        while len(buffer) < buffer_len:
            with torch.no_grad():
                decisions = agent(worlds, value=True)
            new_worlds, transition = worlds.step(decisions.actions)
            buffer.append(arrdict.arrdict(worlds=worlds, decisions=decisions.half(),
                                          transitions=learning.half(transition)).detach())
            worlds = new_worlds
        chunk, buffer = as_chunk(buffer, n_envs)
        pl, vl = optimize(network, scaler, opt, chunk[idxs])
        if on_step is not None:
            on_step(step, pl, vl)
    return worlds

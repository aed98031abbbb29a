"""The actor/learner loop around the search (boardlaw/main.py:61-98,147-200), without the reference's run directory,
stats, storage and live-arena side effects: collect `buffer_len` moves of self-play, turn them into a chunk with
reward-to-go targets, take one AMP Adam step on one random timestep per env.

Multi-GPU: one process per GPU, each with its own env shard; the only collective is the gradient all-reduce
(`parallel.allreduce_gradients`: one flat bucket over RCCL -- the 512x4 network is 1.2 M parameters, a single 4.7 MB
fp32 bucket, latency- not bandwidth-bound on xGMI's point-to-point links, so bucketing finer would only add launches).
The reference is single-device; with world size 1 this is its `optimize`."""
import time

import numpy as np
import torch

from . import arrdict, learning, parallel


def _sync(dev):
    if dev.type == 'cuda':
        torch.cuda.synchronize(dev)


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


def optimize(network, scaler, opt, batch, sync_gradients=True, bucket=None):
    """One learner step (main.py:76-98).  Returns (policy_loss, value_loss) as detached tensors.
    bucket: a parallel.GradientBucket over `network` -- the gradients then live in one flat buffer that is zeroed, accumulated
    into, all-reduced and read by the optimiser in place (what `run` uses when there are several ranks)."""
    cuda = next(network.parameters()).is_cuda
    with torch.autocast('cuda', enabled=cuda):
        policy_loss, value_loss = losses(network, batch)
        loss = policy_loss + value_loss
    if bucket is not None:
        bucket.zero()
    else:
        opt.zero_grad()
    scaler.scale(loss).backward()
    if sync_gradients:
        if bucket is not None:
            bucket.allreduce()
        else:
            parallel.allreduce_gradients(network)
    scaler.step(opt)
    scaler.update()
    return policy_loss.detach(), value_loss.detach()


def run(worlds, network, n_steps, nodes=64, c_puct=1 / 16, lr=1e-3, buffer_len=64, graph=False, inference=None, on_step=None, timings=None):
    """main.py:147-200 minus run bookkeeping.  `worlds` is this rank's env shard.  Returns the final worlds.
    inference: None (the module under autocast, as the reference), 'torch' or 'fused' (networks.Inference plans; their
    f16 weights are refreshed from the module at the start of every move, so optimiser steps are seen).

    `worlds` may also be a list of independent env batches -- several ACTORS on this GPU sharing the network.  Each actor
    searches on its own stream with its own generator (the search kernels are latency-bound, so a second resident search
    fills the cycles the first leaves idle: DESIGN.md section 5), keeps its own buffer, and the learner takes one step per
    actor chunk in turn -- what that many reference processes sharing one set of weights would do.  Returns the list."""
    from . import networks
    from .mcts import MCTSAgent, MoveRng
    many = isinstance(worlds, (list, tuple))
    batches = list(worlds) if many else [worlds]
    dev = batches[0].device
    actor = network if inference is None else networks.Inference(network, fused=(inference == 'fused'))
    concurrent = many and len(batches) > 1 and dev.type == 'cuda'
    agents, streams = [], []
    for i in range(len(batches)):
        kwargs = {}
        if concurrent:
            gen = torch.Generator(device=dev)
            gen.manual_seed(int(torch.randint(2 ** 31 - 1, (1,)).item()))      # seeded from the default generator: reproducible
            kwargs['rng'] = MoveRng(generator=gen)
            streams.append(torch.cuda.Stream(device=dev))
        agents.append(MCTSAgent(actor, n_nodes=nodes, c_puct=c_puct, graph=graph, **kwargs))
    opt = torch.optim.Adam(network.parameters(), lr=lr)
    scaler = torch.amp.GradScaler('cuda', enabled=(dev.type == 'cuda'))
    # several ranks: the gradients live in one flat buffer that is all-reduced where it lies (parallel.GradientBucket)
    # (built ONCE; under the benchmark the same code path runs in a one-rank group, too, every collective bracketed by device events)
    bucket = None
    if torch.distributed.is_initialized() and (timings is not None or torch.distributed.get_world_size() > 1):
        bucket = parallel.GradientBucket(network, always=timings is not None, timed=timings is not None)
        if timings is not None:
            timings['bucket'] = bucket
    idxs = [(torch.randint(buffer_len, (w.n_envs,), device=dev), torch.arange(w.n_envs, device=dev)) for w in batches]
    buffers = [[] for _ in batches]

    def move(i):
        with torch.no_grad():
            decisions = agents[i](batches[i], value=True)
        new_worlds, transition = batches[i].step(decisions.actions)
        buffers[i].append(arrdict.arrdict(worlds=batches[i], decisions=decisions.half(),
                                          transitions=learning.half(transition)).detach())
        batches[i] = new_worlds

    def refresh_actor():
        # all actors share ONE inference plan whose static f16 / packed weight buffers are rewritten in place: do it here, on
        # the learner's stream, BEFORE the actors' streams wait on that stream -- otherwise the first actor to move refreshes on
        # its own stream while another replays a captured move that reads the same buffers
        if hasattr(actor, 'refresh_if_stale'):
            actor.refresh_if_stale()

    if concurrent:
        refresh_actor()
        for s_ in streams:
            s_.wait_stream(torch.cuda.current_stream(dev))
    for step in range(n_steps):
        if timings is not None:
            _sync(dev); t_play = time.perf_counter(); moves_before = sum(len(b) for b in buffers)
        while any(len(b) < buffer_len for b in buffers):
            for i in range(len(batches)):
                if len(buffers[i]) < buffer_len:
                    if concurrent:
                        with torch.cuda.stream(streams[i]):
                            move(i)
                    else:
                        move(i)
        if concurrent:
            for s_ in streams:                                  # the learner reads what the actors' streams produced ...
                torch.cuda.current_stream(dev).wait_stream(s_)
        if timings is not None:
            _sync(dev)
            timings.setdefault('selfplay_s', []).append(time.perf_counter() - t_play)
            timings.setdefault('moves', []).append(sum(len(b) for b in buffers) - moves_before)
        for i in range(len(batches)):
            if timings is not None:
                _sync(dev); t0 = time.perf_counter()
            chunk, buffers[i] = as_chunk(buffers[i], batches[i].n_envs)
            pl, vl = optimize(network, scaler, opt, chunk[idxs[i]], bucket=bucket)
            if timings is not None:
                _sync(dev); timings.setdefault('learner_s', []).append(time.perf_counter() - t0)
            if on_step is not None:
                on_step(step, pl, vl)
        if concurrent:
            refresh_actor()
            for s_ in streams:                                  # ... and the next moves see the updated weights (and freed blocks)
                s_.wait_stream(torch.cuda.current_stream(dev))
    return batches if many else batches[0]

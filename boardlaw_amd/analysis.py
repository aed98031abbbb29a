"""Self-play traces: `rollout` and the per-agent decision record (the role of boardlaw/analysis.py:11-87).

`rollout(worlds, agents, n_steps= | n_trajs= | n_reps=)` lets agent i act in the envs where seat i is to move
(variable batch size per call), steps the whole batch, and returns everything stacked over time:

    trace.actions      (T,B)         the merged actions
    trace.transitions  (T,B,...)     what worlds.step returned
    trace.worlds       (T,B,...)     the worlds AFTER each step
    trace.decisions    {'0': ..., '1': ...}   per agent, every field of its decisions widened to all B envs --
                       NaN (floating) / -1 (integer) where the agent did not act -- plus `mask` (T,B) saying where it did
"""
import torch

from . import arrdict


def combine_actions(decisions, masks):
    """One (B,) action tensor from per-seat decisions over disjoint env masks (analysis.py:11-15)."""
    merged = torch.cat([d.actions for d in decisions.values()])
    for seat, d in decisions.items():
        merged[masks[seat]] = d.actions
    return merged


def _blank_like(exemplar, n_envs):
    if exemplar.dtype.is_floating_point:
        fill = float('nan')
    elif exemplar.dtype in (torch.short, torch.int, torch.long):
        fill = -1
    else:
        raise ValueError(f'Don\'t have a default for "{exemplar.dtype}"')
    return torch.full((n_envs, *exemplar.shape[1:]), fill, dtype=exemplar.dtype, device=exemplar.device)


def combine_decisions(dtrace, mtrace):
    """dtrace[t] = {agent: decisions over the envs it moved in}, mtrace[t] = {agent: (B,) bool mask} ->
    {str(agent): decisions stacked over t and widened to all B envs, with `mask`} (analysis.py:28-45)."""
    n_envs = next(iter(mtrace[0].values())).shape[0]
    out = arrdict.arrdict()
    for agent in sorted({a for step in dtrace for a in step}):
        exemplar = next(step[agent] for step in dtrace if agent in step)
        device = next(iter(arrdict.leaves(exemplar))).device
        frames = []
        for step, masks in zip(dtrace, mtrace):
            frame = exemplar.map(_blank_like, n_envs=n_envs)
            if agent in masks:
                frame[masks[agent]] = step[agent]
                frame['mask'] = masks[agent]
            else:
                frame['mask'] = torch.zeros((n_envs,), dtype=torch.bool, device=device)
            frames.append(frame)
        out[str(agent)] = arrdict.stack(frames)
    return out


@torch.no_grad()
def rollout(worlds, agents, n_steps=None, n_trajs=None, n_reps=None, **kwargs):
    """Stops after n_steps steps, n_trajs finished games in total, or n_reps finished games in every env
    (analysis.py:47-87)."""
    if sum(x is not None for x in (n_steps, n_trajs, n_reps)) != 1:
        raise AssertionError('Must specify exactly one of n_steps or n_trajs or n_reps')
    frames, dtrace, mtrace = [], [], []
    finished, per_env = 0, torch.zeros(worlds.n_envs, device=worlds.device)
    while True:
        per_seat, masks = {}, {}
        for seat, agent in enumerate(agents):
            here = worlds.seats == seat
            if bool(here.any()):
                per_seat[seat], masks[seat] = agent(worlds[here], **kwargs), here
        actions = combine_actions(per_seat, masks)
        worlds, transitions = worlds.step(actions)
        frames.append(arrdict.arrdict(actions=actions, transitions=transitions, worlds=worlds))
        dtrace.append(per_seat)
        mtrace.append(masks)
        finished += int(transitions.terminal.sum())
        per_env += transitions.terminal
        if (n_steps and len(frames) >= n_steps) or (n_trajs and finished >= n_trajs) or (n_reps and bool((per_env >= n_reps).all())):
            break
    trace = arrdict.stack(frames)
    trace['decisions'] = combine_decisions(dtrace, mtrace)
    return trace

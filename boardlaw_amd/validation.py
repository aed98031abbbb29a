"""Toy worlds with planted policies/values and the agents that read them: fixtures for the search tests
(the roles of boardlaw/validation.py:16-77,82-278).  Each world follows the world protocol the search relies on:
`n_envs, n_seats, device, action_space, valid, seats, step(actions) -> (world, arrdict(terminal, rewards))`."""
import torch

from . import arrdict, heads


class ProxyAgent:
    """Returns the logits/values the world itself carries."""

    def __call__(self, world, value=False, eval=False):
        return arrdict.arrdict(logits=world.logits, v=world.v)


class RandomAgent:
    """Uniform over valid actions, value 0."""

    def __call__(self, world, value=True, eval=False):
        valid = world.valid
        return arrdict.arrdict(
            logits=torch.log(valid.float() / valid.sum(-1, keepdims=True)),
            actions=torch.distributions.Categorical(probs=valid.float()).sample(),
            v=torch.zeros((valid.shape[0], world.n_seats), device=world.device))


def _categorical(probs=None, logits=None):
    """The reference's draw: torch.distributions.Categorical(...).sample() from torch's default generator."""
    return torch.distributions.Categorical(probs=probs, logits=logits).sample()


class MonteCarloAgent:
    """Flat Monte-Carlo evaluation (boardlaw/validation.py:32-77): `n_rollouts` uniformly random playouts from the given worlds,
    each credited to the playout's first action; logits = log-softmax(temperature x mean return of the mover per first action),
    -inf on invalid actions; v = mean return per seat over all playouts.

    Random draws, in the reference's order: per rollout one uniform draw over the valid actions per step until every env has
    finished a game, then one draw from the final logits.  `sampler(probs=..., logits=...)` makes them (default: torch's
    Categorical from the default generator, i.e. the reference's stream; tests replay recorded draws through it).

    One quirk of the reference is kept because results depend on it: the tallies are indexed `[envs, a[:, None]]`, which
    broadcasts to (B, B) -- every env's tally row receives the returns of EVERY env's playout (at its first action), written
    without accumulation (where two envs share a first action one of them wins: the last on the CPU).  With one env it is the
    textbook estimator; with several, every env ends up with the same v."""

    def __init__(self, n_rollouts, temperature=1., sampler=None):
        self.n_rollouts, self.temperature = n_rollouts, temperature
        self.sampler = sampler or _categorical

    def rollout(self, world):
        """One playout per env -> (return per seat summed until the env's first terminal step (B,S), first actions (B,))."""
        n_envs = world.valid.shape[0]
        alive = torch.ones((n_envs,), dtype=torch.bool, device=world.device)
        returns = torch.zeros((n_envs, world.n_seats), dtype=torch.float, device=world.device)
        first = None
        while bool(alive.any()):
            actions = self.sampler(probs=world.valid.float())
            first = actions if first is None else first
            world, transition = world.step(actions)
            returns = returns + transition.rewards * alive[:, None].float()
            alive = alive & ~transition.terminal
        return returns, first

    def __call__(self, world, value=True, eval=False):
        envs = torch.arange(world.n_envs, device=world.device)
        totals = torch.zeros(tuple(world.valid.shape) + (world.n_seats,), dtype=torch.float, device=world.device)      # (B, A, S)
        counts = torch.zeros_like(totals)
        for _ in range(self.n_rollouts):
            returns, first = self.rollout(world)
            where = (envs, first[:, None])                      # broadcasts to (B, B): see the class docstring
            totals[where] = totals[where] + returns[:, None]
            counts[where] = counts[where] + 1.
        means = torch.where(counts > 0, totals / counts, torch.zeros_like(counts))
        mover = means[envs, :, world.seats.long()]              # (B, A): the mover's mean return per first action
        logits = torch.log_softmax(self.temperature * mover, -1)
        logits[~world.valid] = -float('inf')
        return arrdict.arrdict(logits=logits, actions=self.sampler(logits=logits), v=totals.sum(-2) / counts.sum(-2))


def uniform_logits(valid):
    return torch.log(valid.float() / valid.sum(-1, keepdims=True))


class Win(arrdict.namedarrtuple('Win', fields=('envs',))):
    """One seat, one action, every step terminal with reward +1; planted v = 1."""

    @classmethod
    def initial(cls, n_envs=1, device='cuda'):
        return cls(envs=torch.arange(n_envs, device=device))

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        envs = self['envs']
        if not isinstance(envs, torch.Tensor):
            return
        self.device, self.n_envs, self.n_seats = envs.device, len(envs), 1
        self.obs_space, self.action_space = (0,), (1,)
        self.valid = torch.ones_like(envs[..., None], dtype=torch.bool)
        self.seats = torch.zeros_like(envs)
        self.logits = uniform_logits(self.valid)
        self.v = torch.ones_like(self.valid, dtype=torch.float)

    def step(self, actions):
        envs = self['envs']
        return self, arrdict.arrdict(terminal=torch.ones_like(envs, dtype=torch.bool),
                                     rewards=torch.ones_like(envs, dtype=torch.float)[..., None])


class WinnerLoser(arrdict.namedarrtuple('WinnerLoser', fields=('seats',))):
    """Two seats alternate; when seat 1 moves the episode ends with rewards (+1, -1); planted v = (+1, -1)."""

    @classmethod
    def initial(cls, n_envs=1, device='cuda'):
        return cls(seats=torch.zeros(n_envs, device=device, dtype=torch.int))

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        seats = self['seats']
        if not isinstance(seats, torch.Tensor):
            return
        self.device, self.n_envs, self.n_seats = seats.device, seats.shape[0], 2
        self.obs_space, self.action_space = (0,), (1,)
        self.valid = torch.ones(seats.shape + (1,), dtype=torch.bool, device=self.device)
        self.logits = uniform_logits(self.valid)
        self.v = torch.stack([torch.ones_like(seats), -torch.ones_like(seats)], -1).float()

    def step(self, actions):
        seats = self['seats']
        terminal = seats == 1
        return type(self)(seats=1 - seats), arrdict.arrdict(
            terminal=terminal, rewards=torch.stack([terminal.float(), -terminal.float()], -1))


class All(arrdict.namedarrtuple('All', fields=('history', 'count'))):
    """Each seat must play action 1 on every one of `length` turns to earn +1; planted v is the exact value under a
    uniform policy: 2^-(turns left) while still on track, else 0."""

    @classmethod
    def initial(cls, n_envs=1, n_seats=1, length=4, device='cuda'):
        return cls(history=torch.full((n_envs, length, n_seats), -1, dtype=torch.long, device=device),
                   count=torch.zeros((n_envs,), dtype=torch.long, device=device))

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        history, count = self['history'], self['count']
        if not isinstance(count, torch.Tensor):
            return
        self.n_envs, self.length, self.n_seats = history.shape[0], history.shape[-2], history.shape[-1]
        self.device = count.device
        self.max_count = self.n_seats * self.length
        self.obs_space, self.action_space = heads.Tensor((1,)), heads.Masked(2)
        self.valid = torch.ones(count.shape + (2,), dtype=torch.bool, device=self.device)
        self.seats = count % self.n_seats
        self.obs = count[..., None].float() / self.max_count
        self.envs = torch.arange(self.n_envs, device=self.device)
        self.logits = uniform_logits(self.valid)
        ones = (history == 1).sum(-2)
        on_track = ones == count[..., None]
        self.v = on_track.float() * 2**((ones - self.length).float())

    def step(self, actions):
        history, count = self['history'].clone(), self['count']
        history[self.envs, count // self.n_seats, self.seats] = actions
        count = count + 1
        terminal = count == self.max_count
        rewards = (terminal[:, None] & (history == 1).all(-2)).float()
        count[terminal] = 0
        history[terminal] = -1
        return type(self)(history=history, count=count), arrdict.arrdict(terminal=terminal, rewards=rewards)


class SequentialMatrix(arrdict.namedarrtuple('SequentialMatrix', fields=('payoffs', 'moves', 'seats'))):
    """A 2x2 matrix game played in sequence (boardlaw/validation.py:213-278): seat 0 picks a row, seat 1 -- who sees it: obs is
    seat 0's move -- picks a column, then both are paid `payoffs[row, col]` (a pair) and the episode restarts.
    payoffs (B,2,2,2) f32, moves (B,2) i32 = -1 until made, seats (B) i32.  The reference's per-outcome counters
    (`stats.mean('outcomes/i-j', ...)`, its run-statistics layer) are out of scope: `outcome_hook(moves_of_finished_envs)` is
    called in their place when set."""

    outcome_hook = None

    @classmethod
    def initial(cls, payoff, n_envs=1, device='cuda'):
        table = torch.as_tensor(payoff, dtype=torch.float).to(device)
        return cls(payoffs=table[None].repeat(n_envs, 1, 1, 1),
                   moves=torch.full((n_envs, 2), -1, dtype=torch.int, device=device),
                   seats=torch.zeros((n_envs,), dtype=torch.int, device=device))

    @classmethod
    def dilemma(cls, *args, **kwargs):
        return cls.initial([[[0., 0.], [1., 0.]],
                            [[0., 1.], [.5, .5]]], *args, **kwargs)

    @classmethod
    def antisymmetric(cls, *args, **kwargs):
        return cls.initial([[[1., 0.], [1., 1.]],
                            [[0., 0.], [0., .1]]], *args, **kwargs)

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        payoffs, moves, seats = self['payoffs'], self['moves'], self['seats']
        if not isinstance(payoffs, torch.Tensor):
            return
        self.n_envs, self.n_seats, self.device = seats.shape[-1], 2, seats.device
        self.obs_space, self.action_space = heads.Tensor((1,)), heads.Masked(2)
        self.obs = moves[..., :1].float()
        self.valid = torch.ones(seats.shape + (2,), dtype=torch.bool, device=self.device)
        self.envs = torch.arange(self.n_envs, device=self.device)

    def step(self, actions):
        payoffs, seats = self['payoffs'], self['seats']
        moves = self['moves'].clone()
        moves[self.envs, seats.long()] = actions.int()
        over = seats == 1                                        # seat 1 has just moved: pay out and restart
        if type(self).outcome_hook is not None and bool(over.any()):
            type(self).outcome_hook(moves[over])
        rewards = torch.zeros_like(payoffs[:, 0, 0])
        rewards[over] = payoffs[self.envs[over], moves[over, 0].long(), moves[over, 1].long()]
        next_seats = torch.where(over, torch.zeros_like(seats), seats + 1)
        moves[over] = -1
        return type(self)(payoffs=payoffs, moves=moves, seats=next_seats), arrdict.arrdict(terminal=over, rewards=rewards)

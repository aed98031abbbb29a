"""Toy worlds with planted policies/values and the agents that read them: fixtures for the search tests
(the roles of boardlaw/validation.py:16-21,82-200).  Each world follows the world protocol the search relies on:
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

"""The Hex world: batched boards + seats with lazy `obs`/`valid` and `step` (boardlaw/hex/__init__.py:129-195).

Layout contract (unchanged from the reference): `board` (B,S,S) u8 with cell codes `.bwTBLR` = 0..6, `seats` (B,) i32,
`obs` (B,S,S,2) f32 in the mover's frame, `valid` (B,S*S) bool, `step(actions) -> (Hex, arrdict(terminal, rewards))`."""
import torch

from .. import arrdict, heads
from . import cuda

CHARS = '.bwTBLR'
ORDS = {c: i for i, c in enumerate(CHARS)}


class Hex(arrdict.namedarrtuple('Hex', fields=('board', 'seats'))):

    @classmethod
    def initial(cls, n_envs, boardsize=11, device='cuda'):
        # black (seat 0) moves first, hex/__init__.py:131-136
        return cls(board=torch.zeros((n_envs, boardsize, boardsize), device=device, dtype=torch.uint8),
                   seats=torch.zeros((n_envs,), device=device, dtype=torch.int))

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if not isinstance(self['board'], torch.Tensor):
            return  # an intermediate dict of bound methods produced by attribute delegation (e.g. `.clone`)
        board = self['board']
        self.n_seats = 2
        self.n_envs = board.shape[0]
        self.boardsize = board.shape[-1]
        self.device = board.device
        self.obs_space = heads.Tensor((self.boardsize, self.boardsize, 2))
        self.action_space = heads.Masked(self.boardsize * self.boardsize)
        self._obs = None
        self._valid = None

    def _observe(self):
        b, s = self.board, self.seats
        if b.ndim == 3 and s.dtype == torch.int32 and b.dtype == torch.uint8 and b.is_contiguous() and s.is_contiguous() and b.is_cuda:
            self._obs, self._valid = cuda.observe_valid(b, s)          # both in one launch
        else:
            self._obs = cuda.observe(b, s)

    @property
    def obs(self):
        if self._obs is None:
            self._observe()
        return self._obs

    @property
    def valid(self):
        if self._valid is None:
            if self._obs is None:
                self._observe()
            if self._valid is None:
                lead = self.board.shape[:-2]
                self._valid = (self.obs == 0).all(-1).reshape(*lead, -1)
        return self._valid

    def step(self, actions, reset=True, check=True):
        """actions: (B,) flat cell indices in the mover's frame, or (B,2) (row, col).  hex/__init__.py:161-195.
        `check=False` skips the reference's three validity asserts (each one a device->host sync)."""
        if self.board.ndim != 3:
            raise ValueError('You can only step a board with a single batch dimension')
        if check:
            assert (0 <= actions).all(), 'You passed a negative action'
        if actions.ndim == 2:
            actions = actions[..., 0] * self.boardsize + actions[:, 1]
        assert actions.shape == (self.n_envs,)
        if check:
            assert self.valid.gather(1, actions[:, None].long()).squeeze(-1).all()

        if reset and self.seats.dtype == torch.int32 and self.board.is_contiguous() and self.seats.is_contiguous():
            # the whole transition as one launch (bl_hex_world_step); same results as the steps below
            new_board, new_seats, rewards, terminal = cuda.world_step(self.board, self.seats, actions)
            return type(self)(board=new_board, seats=new_seats), arrdict.arrdict(terminal=terminal, rewards=rewards)
        new_board = self.board.clone()
        rewards = cuda.step(new_board, self.seats.int(), actions.int())
        if reset:
            terminal = (rewards > 0).any(-1)
        else:
            terminal = torch.zeros((self.n_envs,), dtype=torch.bool, device=self.device)
        new_board[terminal] = 0
        new_seats = 1 - self.seats
        new_seats[terminal] = 0
        return type(self)(board=new_board, seats=new_seats), arrdict.arrdict(terminal=terminal, rewards=rewards)


class Solitaire(Hex):
    """One-player Hex (hex/__init__.py:224-255): seat 0 is the player; after its move a scripted opponent (`_play`)
    answers in every env where it is now seat 1's turn, and the player sees the summed transition for its own seat
    only -- rewards (B,1)."""

    @classmethod
    def initial(cls, *args, seat=0, **kwargs):
        if seat == 1:
            raise ValueError('Can\'t do seat #1 right now')
        return super().initial(*args, **kwargs)

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if isinstance(self['board'], torch.Tensor):
            self.n_seats = 1

    def step(self, actions, **kwargs):
        worlds, transitions = super().step(actions, **kwargs)
        # a winning move resets the env to seat 0 already; everywhere else the opponent answers until the player is
        # to move again (one answer, unless that answer itself ends the game -- then the reset hands the move back)
        while True:
            theirs = worlds.seats != self.seats
            if not bool(theirs.any()):
                break
            worlds[theirs], reply = self._play(worlds[theirs])
            transitions.rewards[theirs] += reply.rewards
            transitions.terminal[theirs] |= reply.terminal
        mine = self.seats.long()[:, None]
        transitions['rewards'] = transitions.rewards.gather(1, mine)
        return worlds, transitions


class Lazy(Solitaire):
    """Opponent plays the first available cell (hex/__init__.py:257-266)."""

    @classmethod
    def _reply(cls, worlds):
        return worlds.valid.int().argmax(-1)        # lowest index with valid == True

    @classmethod
    def _play(cls, worlds):
        return Hex.step(worlds, cls._reply(worlds))


class Random(Solitaire):
    """Opponent plays a uniformly random available cell (hex/__init__.py:268-274)."""

    @classmethod
    def _reply(cls, worlds):
        return torch.distributions.Categorical(probs=worlds.valid.float()).sample()

    @classmethod
    def _play(cls, worlds):
        return Hex.step(worlds, cls._reply(worlds))


def board_actions(s):
    """Move list that reproduces a board drawn with `b`/`w`/`.` rows (boardlaw/hex/tests.py:101-122)."""
    rows = [l.strip() for l in s.splitlines() if l.strip()]
    blacks = [(i, j) for i, r in enumerate(rows) for j, c in enumerate(r) if c == 'b']
    whites = [(i, j) for i, r in enumerate(rows) for j, c in enumerate(r) if c == 'w']
    assert len(blacks) - len(whites) in (0, 1)
    moves = []
    for k in range(len(whites)):
        moves.append(list(blacks[k]))
        moves.append([whites[k][1], whites[k][0]])     # white acts in the transposed frame
    if len(whites) < len(blacks):
        moves.append(list(blacks[-1]))
    return torch.tensor(moves, dtype=torch.long).reshape(-1, 2), len(rows)


def from_string(s, **kwargs):
    moves, size = board_actions(s)
    worlds = Hex.initial(n_envs=1, boardsize=size, **kwargs)
    for a in moves.to(worlds.device):
        worlds, _ = worlds.step(a[None])
    return worlds

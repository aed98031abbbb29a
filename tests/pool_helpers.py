"""Module-level (picklable) world / agent factories for the arena fan-out tests: the workers of arena.run_jobs import them by
name.  TEST INFRASTRUCTURE: the CPU world is the oracle-backed Hex stand-in of test_reference_fixtures."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
for p in (HERE, os.path.dirname(HERE)):
    if p not in sys.path:
        sys.path.insert(0, p)


def cpu_worlds(n_envs, S=5):
    import oracle_lib
    from test_reference_fixtures import oracle_hex
    return oracle_hex(oracle_lib.load()).initial(n_envs, S, device='cpu')


def gpu_worlds(n_envs, S=5):
    from boardlaw_amd.hex import Hex
    return Hex.initial(n_envs, S)


def edge_agent(name):
    from test_reference_fixtures import EdgeAgent
    k = int(name[1:])
    return EdgeAgent(from_end=bool(k % 2), k=k // 2)


def square(x):
    return x * x


def die(code):
    """A worker that ends without posting a result -- what a crash in the native library or an OOM kill looks like to the pool."""
    os._exit(code)


def device_of_worker(_):
    import torch
    return torch.cuda.current_device() if torch.cuda.is_available() else -1


class Accumulator:
    """A MatchPool player that remembers what it has been asked before: persistent workers keep their state between plays."""

    def __init__(self, offset):
        self.offset, self.seen = offset, []

    def __call__(self, job):
        self.seen.append(job)
        return (os.getpid(), job * job + self.offset, len(self.seen))

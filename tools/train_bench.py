"""Config 4 as N ranks: 13x13 Hex, 1024 envs per rank, 256 sims per move, FCModel 1024x8 -- self-play + one learner step per
`buffer` moves, the gradients averaged over ranks by ONE RCCL all-reduce of a persistent flat bucket (parallel.GradientBucket).

    python tools/train_bench.py [--gpus N] [--steps K] [--envs E] [--boardsize S] [--nodes T] [--width W] [--depth D] [--buffer L]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... tools/train_bench.py --gpus N

One process per GPU (the reference's model, boardlaw/main.py:202-209): `--gpus N` without a launcher re-executes itself under
torch.distributed.run on 127.0.0.1.  Prints ONE JSON line from rank 0: self-play ms per move, learner-step ms (as_chunk +
forward/backward/all-reduce/Adam), all-reduce ms (device events around the collective + its scaling), sims/s of the whole job,
`ranks_seen` through the collective itself.  TRAIN_DRY=1: the same skeleton on the CPU over gloo with a tiny shape and the
generic search path replaced by a stand-in (no GPU work) -- what the CPU test runs for N = 2."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def respawn(args):
    world = os.environ.get('WORLD_SIZE')
    if world is not None:
        if int(world) != args.gpus:
            raise SystemExit(f'train_bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks')
        return
    if args.gpus > 1:
        import socket
        with socket.socket() as sock:
            sock.bind(('127.0.0.1', 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
               '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        os.execv(sys.executable, cmd)


def dry_run(args):
    """The rendezvous, the bucket all-reduce and the report on CPU ranks over gloo: a small FCModel, synthetic gradients."""
    from boardlaw_amd import heads, networks, parallel
    rank, world, _ = parallel.env_rank()
    parallel.init('gloo')
    if world == 1 and not torch.distributed.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
        torch.distributed.init_process_group('gloo', rank=0, world_size=1)
    torch.manual_seed(0)
    net = networks.FCModel(heads.Tensor((5, 5, 2)), heads.Masked(25), width=32, depth=2)
    bucket = parallel.GradientBucket(net, always=True)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    t_ar = []
    for step in range(args.steps):
        bucket.zero()
        for i, p in enumerate(net.parameters()):
            p.grad.add_(float(rank + 1) * (i + 1))             # rank r contributes (r + 1) * (i + 1): the mean is known
        t0 = time.perf_counter(); bucket.allreduce(); t_ar.append(time.perf_counter() - t0)
        mean = sum(range(1, world + 1)) / world
        assert bucket.intact() and all(torch.allclose(p.grad, torch.full_like(p.grad, mean * (i + 1))) for i, p in enumerate(net.parameters()))
        opt.step()
    per_rank, seen = parallel.gather_over_ranks(float(rank + 1), device='cpu')
    if rank == 0:
        print(json.dumps({'metric': 'train_bench', 'dry_run': True, 'n_gpus': world, 'steps': args.steps, 'allreduce_ms': 1e3 * float(np.mean(t_ar)),
                          'bucket_mb': bucket.flat.numel() * 4 / 2**20, 'ranks_seen': seen, 'per_rank_values': per_rank}))
    torch.distributed.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3, help='learner steps (each after `buffer` more moves of self-play... the first after filling the buffer)')
    ap.add_argument('--envs', type=int, default=1024, help='envs per rank (config 4: 8192 over 8 GPUs)')
    ap.add_argument('--boardsize', type=int, default=13)
    ap.add_argument('--nodes', type=int, default=256)
    ap.add_argument('--width', type=int, default=1024)
    ap.add_argument('--depth', type=int, default=8)
    ap.add_argument('--buffer', type=int, default=8, help='moves the buffer holds (the reference: 64); after the first fill every learner step follows ONE new move')
    args = ap.parse_args()
    respawn(args)
    if os.environ.get('TRAIN_DRY') == '1':
        return dry_run(args)
    assert torch.cuda.is_available(), 'train_bench.py needs an MI355X (TRAIN_DRY=1 is the CPU skeleton)'
    from boardlaw_amd import networks, parallel, training
    from boardlaw_amd.hex import Hex
    rank, world, local = parallel.env_rank()
    assert world == args.gpus, (world, args.gpus)
    local = int(os.environ.get('BENCH_FORCE_DEVICE', local))
    torch.cuda.set_device(local)
    backend = os.environ.get('BENCH_BACKEND', 'nccl')
    parallel.init(backend)
    if not torch.distributed.is_initialized():         # one rank: a one-rank group, so that the collective's code path runs and is timed
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29534')
        kw = {'device_id': torch.device('cuda', local)} if backend == 'nccl' else {}
        torch.distributed.init_process_group(backend, rank=0, world_size=1, **kw)
    torch.manual_seed(0)                               # the same initial weights on every rank
    worlds = Hex.initial(args.envs, args.boardsize)
    net = networks.FCModel(worlds.obs_space, worlds.action_space, width=args.width, depth=args.depth).cuda()
    torch.manual_seed(1 + rank)                        # ... and a different random stream per shard
    timings = {}
    parallel.barrier()
    t0 = time.perf_counter()
    training.run(worlds, net, n_steps=args.steps, nodes=args.nodes, buffer_len=args.buffer, graph=True, inference='fused', timings=timings)
    parallel.barrier()
    elapsed = parallel.max_over_ranks(time.perf_counter() - t0)
    moves = int(sum(timings['moves']))
    per_rank, seen = parallel.gather_over_ranks(args.envs * args.nodes * moves / elapsed)
    # every rank ends with the same weights: the gradients were averaged, the initial weights and the optimiser are the same
    digest = torch.stack([p.detach().double().sum() for p in net.parameters()]).sum()
    lo, hi = -parallel.max_over_ranks(-float(digest)), parallel.max_over_ranks(float(digest))
    if rank == 0:
        ar = timings['bucket'].collective_ms()
        print(json.dumps({
            'metric': 'train_bench', 'n_gpus': world, 'ranks_seen': seen, 'per_rank_sims_per_sec': per_rank, 'backend': torch.distributed.get_backend(),
            'config': {'workload': f'{args.boardsize}x{args.boardsize} Hex, {args.envs} envs/rank x {args.nodes} sims/move, FCModel {args.width}x{args.depth}, '
                                   f'buffer of {args.buffer} moves, 1 new move per learner step once it is full (the reference\'s main.py semantics: as_chunk drops batch_size/B = 1 step)' + (' (BASELINE config 4 per-GPU shape)' if (args.boardsize, args.envs, args.nodes, args.width, args.depth) == (13, 1024, 256, 1024, 8) else '')},
            # by learner step: the first carries the moves' graph capture and warm-up, the later ones are the steady rate
            'selfplay_ms_per_move_by_step': [round(1e3 * t / max(m, 1), 3) for t, m in zip(timings['selfplay_s'], timings['moves'])],
            'selfplay_ms_per_move': round(1e3 * timings['selfplay_s'][-1] / max(timings['moves'][-1], 1), 3), 'moves': moves,
            'learner_step_ms': [round(1e3 * x, 3) for x in timings['learner_s']],
            'allreduce_ms': [round(x, 3) for x in ar], 'bucket_mb': round(timings['bucket'].flat.numel() * 4 / 2**20, 2),
            # whole job = including graph capture, warm-up and the first (slow) learner step; steady state = learner steps 1.. only
            'sims_per_sec_whole_job': args.envs * args.nodes * moves * world / elapsed, 'elapsed_s': elapsed,
            'sims_per_sec_steady_state': (args.envs * args.nodes * world * sum(timings['moves'][1:]) / max(sum(timings['selfplay_s'][1:]) + sum(timings['learner_s'][1:]), 1e-9)
                                          if len(timings['moves']) > 1 else None),
            'weights_identical_over_ranks': lo == hi}))
    torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()

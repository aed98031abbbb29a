"""Do independent branches overlap on this stack?  K chains of N small dependent kernels each (a chain = one env group's launch
sequence in miniature: every kernel a few microseconds on a few CUs), issued (a) on one stream, (b) on K streams eagerly,
(c) captured as ONE graph with K forked branches, (d) captured as K graphs replayed on K streams.  Prints microseconds per
round of K chains.  usage: python tools/micro/graph_fork.py [N]"""
import sys, time
import torch

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = 'cuda'


def chain(x, w):
    for _ in range(N):
        x = torch.tanh(x @ w)          # (64, 256) @ (256, 256): a handful of workgroups, ~5-8 us
    return x


def timed(f, reps=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


for K in (1, 2, 4, 8):
    xs = [torch.randn(64, 256, device=dev, dtype=torch.half) for _ in range(K)]
    ws = [torch.randn(256, 256, device=dev, dtype=torch.half) * 0.05 for _ in range(K)]
    streams = [torch.cuda.Stream() for _ in range(K)]
    out = [None] * K

    def one_stream():
        for i in range(K):
            out[i] = chain(xs[i], ws[i])

    def k_streams():
        cur = torch.cuda.current_stream()
        for i in range(K):
            streams[i].wait_stream(cur)
            with torch.cuda.stream(streams[i]):
                out[i] = chain(xs[i], ws[i])
        for i in range(K):
            cur.wait_stream(streams[i])

    a = timed(one_stream)
    b = timed(k_streams)
    # (c) one graph, forked branches
    k_streams(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        k_streams()
    c = timed(g.replay)
    # (d) K graphs on K streams
    gs = []
    for i in range(K):
        gi = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gi):
            out[i] = chain(xs[i], ws[i])
        gs.append(gi)

    def k_graphs():
        for i in range(K):
            with torch.cuda.stream(streams[i]):
                gs[i].replay()
    d = timed(k_graphs)
    # (e) one graph, everything on one stream
    g1 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1):
        one_stream()
    e = timed(g1.replay)
    print(f'K = {K} chains of {N} kernels: one stream eager {a:8.1f} us | K streams eager {b:8.1f} | one graph, K forked branches {c:8.1f} | '
          f'K graphs on K streams {d:8.1f} | one graph, one branch {e:8.1f}', flush=True)

"""Do two HIP streams really run concurrently when one of them never drains?  Chain BIG (GPU-filling GEMMs,
back to back) on stream i, chain SMALL (hundreds of tiny kernels) on stream j, each replayed from its own
single-stream graph; serial time = t_big + t_small, perfect overlap = max()."""
import time
import torch

dev = torch.device("cuda")
a = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
x = torch.randn(64, 1024, device=dev)
NS = 8
streams = [torch.cuda.Stream() for _ in range(NS)]


def big():
    y = a
    for _ in range(40):
        y = (y @ a) * 0.01
    return y


def small():
    y = x
    for _ in range(600):
        y = y * 1.0001
    return y


def capture(fn, s):
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        fn()
    torch.cuda.synchronize()
    return g


def timed(f, n=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


gb = [capture(big, s) for s in streams]
gs = [capture(small, s) for s in streams]


def on(s, g):
    with torch.cuda.stream(s):
        g.replay()


tb = timed(lambda: on(streams[0], gb[0]))
ts = timed(lambda: on(streams[1], gs[1]))
print("big alone %.2f ms   small alone %.2f ms" % (tb, ts))
for i in range(NS):
    row = []
    for j in range(NS):
        if i == j:
            row.append("  -  ")
            continue
        def both():
            on(streams[i], gb[i])
            on(streams[j], gs[j])
        row.append("%5.2f" % timed(both, 3))
    print("big on s%d | small on s0..s%d: %s" % (i, NS - 1, " ".join(row)), flush=True)

"""How does a replayed HIP graph schedule two independent branches captured from forked streams?
Branch A on a side stream, branch B on the capture stream, each `n` small-grid kernels (they cannot fill the
chip, so real concurrency halves the time).  Variants differ only in the ORDER the branches are issued."""
import sys, time
import torch

dev = torch.device("cuda")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
xa = torch.randn(64, 4096, device=dev)
xb = torch.randn(64, 4096, device=dev)


def work(x):  # ~one wave of blocks, tens of us
    return torch.cumsum(x, 1) * 0.5


def chain(x, k):
    for _ in range(k):
        x = work(x)
    return x


PRE = len(sys.argv) > 2 and sys.argv[2] == "pre"
xp = torch.randn(64, 4096, device=dev)


def variant(name, cap, side):
    r = _variant(name, cap, side)
    return work(r) if PRE else r


def _variant(name, cap, side):
    global xa, xb
    main = torch.cuda.current_stream()
    if PRE:  # a pre-fork node both branches depend on
        p = work(xp)
        xa, xb = p + 1.0, p - 1.0
    if name == "serial":
        return chain(xa, n) + chain(xb, n)
    if name == "A_then_B":  # whole side branch issued first, then the main branch
        side.wait_stream(main)
        with torch.cuda.stream(side):
            a = chain(xa, n)
        b = chain(xb, n)
        main.wait_stream(side)
        return a + b
    if name == "B_then_A":
        side.wait_stream(main)
        b = chain(xb, n)
        with torch.cuda.stream(side):
            a = chain(xa, n)
        main.wait_stream(side)
        return a + b
    if name == "heads_first":  # first kernel of each branch, then the bulk
        side.wait_stream(main)
        with torch.cuda.stream(side):
            a = work(xa)
        b = work(xb)
        with torch.cuda.stream(side):
            a = chain(a, n - 1)
        b = chain(b, n - 1)
        main.wait_stream(side)
        return a + b
    if name == "interleaved":
        side.wait_stream(main)
        a, b = xa, xb
        for _ in range(n):
            with torch.cuda.stream(side):
                a = work(a)
            b = work(b)
        main.wait_stream(side)
        return a + b
    raise KeyError(name)


for name in ["serial", "A_then_B", "B_then_A", "heads_first", "interleaved"]:
    cap, side = torch.cuda.Stream(), torch.cuda.Stream()
    cap.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cap):
        variant(name, cap, side)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(cap):
        for _ in range(5):
            variant(name, cap, side)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 5
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=cap):
        out = variant(name, cap, side)
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    res = []
    for sync_each in (False, True):
        t0 = time.perf_counter()
        for _ in range(10):
            g.replay()
            if sync_each:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / 10)
    print("%-12s eager %.2f ms   replay back-to-back %.2f ms   replay+sync %.2f ms" %
          (name, eager * 1e3, res[0] * 1e3, res[1] * 1e3), flush=True)

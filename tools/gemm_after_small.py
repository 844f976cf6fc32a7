"""Is a big GEMM slower when it follows a run of tiny kernels (the twin K/V projection inside the text chain: 78 us in the step,
45 us back to back)?  Graph-replayed: (k tiny kernels + GEMM) x 24 against (k tiny kernels) x 24 and GEMM x 24.
python tools/gemm_after_small.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bridgeqa_amd import _ext
dev = torch.device("cuda:0")
M, N, K = 16720, 1536, 768
x = (torch.randn(M, K, device=dev)).to(torch.bfloat16)
ws = [(torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16) for _ in range(24)]
b = torch.randn(N, device=dev)
outs = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(2)]
small = torch.randn(320, 768, device=dev)
s = torch.cuda.Stream()


def graph_ms(fn, reps=5):
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        fn()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(s):
            e0.record(); g.replay(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


def tiny(k):
    t = small
    for _ in range(k):
        t = t * 1.0001
    return t


for k in (0, 12, 40):
    both = graph_ms(lambda: [(_ext.gemm_fwd(x, ws[i], b, tile=128, out=outs[i & 1]), tiny(k)) for i in range(24)])
    only_small = graph_ms(lambda: [tiny(k) for i in range(24)]) if k else 0.0
    print("k=%2d tiny kernels between GEMMs: GEMM %.1f us each (chain %.3f ms, tiny alone %.3f ms)"
          % (k, (both - only_small) / 24 * 1e3, both, only_small))

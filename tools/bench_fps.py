"""Micro-benchmark of the FPS kernels on the GPU box (HIP-event timed).  python tools/bench_fps.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bridgeqa_amd import _ext

def scene(B, N, seed=42):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(B, N, 3, generator=g) * torch.tensor([8.0, 8.0, 3.0])).contiguous().cuda()

def timeit(f, n=5):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

for B, N, m in ((16, 40000, 2048), (16, 40000, 1024), (16, 40000, 2), (1, 40000, 2048), (16, 80000, 2048),
                (16, 2048, 1024), (16, 1024, 512), (16, 512, 256), (16, 1024, 256), (16, 8192, 1024), (16, 20000, 2048)):
    x = scene(B, N)
    t = timeit(lambda: _ext.furthest_point_sampling(x, m))
    tb = timeit(lambda: _ext.furthest_point_sampling_bruteforce(x, m), 2) if N > 4096 else float("nan")
    same = torch.equal(_ext.furthest_point_sampling(x, m), _ext.furthest_point_sampling_bruteforce(x, m))
    print("B=%2d N=%6d m=%5d  fps %8.3f ms (%.3f us/round)  bruteforce %8.3f ms  equal=%s  alg %.1f GB/s"
          % (B, N, m, t, t * 1e3 / max(m - 1, 1), tb, same, 20.0 * N * (m - 1) * B / (t * 1e-3) / 1e9))

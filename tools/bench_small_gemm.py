"""forward (K-contiguous P) vs dX (contraction-major P) form of the small-M GEMM kernel on text-side shapes, in isolation
(HIP graph of 50 launches each, so launch gaps are the graph's)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bridgeqa_amd import _ext

def timeit(f, n=50):
    f(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        f()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): f()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

for M, N, K in ((640, 768, 768), (640, 2304, 768), (640, 3072, 768), (640, 768, 3072), (160, 768, 768), (160, 768, 3072),
                (16400, 768, 768), (16400, 2304, 768), (16400, 3072, 768), (16400, 768, 3072)):
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = torch.randn(N, K, device="cuda").bfloat16()
    wt = w.t().contiguous()
    dy = torch.randn(M, N, device="cuda").bfloat16()
    b = torch.randn(N, device="cuda")
    t_f = timeit(lambda: _ext.gemm_fwd(x, w, b))
    t_dx = timeit(lambda: _ext.gemm_dx(dy, w))
    t_dxT = timeit(lambda: _ext.gemm_fwd(dy, wt))          # the same product with a pre-transposed weight (K-contiguous form)
    t_tr = timeit(lambda: w.t().contiguous())
    print("M=%d N=%d K=%d: fwd %.1f us   dX (contraction-major W) %.1f us   dX via W^T (K-contiguous) %.1f us   [transpose of W %.1f us]"
          % (M, N, K, t_f, t_dx, t_dxT, t_tr))

"""a few launches of the small-M GEMM kernel in its forward (K-contiguous W) and dX (contraction-major W) forms on the fc1-dX
shape of the text side (M = 640, contraction 3072, 768 outputs) for rocprofv3 --pmc passes"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bridgeqa_amd import _ext
M, N, K = 640, 3072, 768
w = torch.randn(N, K, device="cuda").bfloat16()
wt = w.t().contiguous()
dy = torch.randn(M, N, device="cuda").bfloat16()
for _ in range(5):
    _ext.gemm_dx(dy, w)        # gemm64_kernel<32, true, false, 0, false>
    _ext.gemm_fwd(dy, wt)      # gemm64_kernel<32, false, false, 0, false>: the same product from a pre-transposed weight
torch.cuda.synchronize()
print("done")

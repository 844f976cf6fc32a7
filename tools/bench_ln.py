"""add+LayerNorm kernels at the ViT residual-stream shape (M = 16 x 1025 rows, H = 768)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bridgeqa_amd import _ext
M, H = 16 * 1025, 768
x = torch.randn(M, H, device="cuda").to(torch.bfloat16); res = torch.randn_like(x); dy = torch.randn_like(x); ds = torch.randn_like(x)
g = torch.ones(H, device="cuda"); b = torch.zeros(H, device="cuda")
def timeit(f, n=20):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
y, s_, mean, rstd, dgb = _ext.drop_add_ln_fwd(x, res, g, b, 1e-6, 0.0, 1, None, True, 0.0, 1025, True)
tf = timeit(lambda: _ext.drop_add_ln_fwd(x, res, g, b, 1e-6, 0.0, 1, None, True, 0.0, 1025, True))
tb = timeit(lambda: _ext.drop_add_ln_bwd(x, res, g, dy, mean, rstd, 1e-6, 0.0, 1, None, ds, 0.0, 1025, torch.zeros(2, H, device="cuda")))
mb = M * H * 2 / 1e6
print("fwd %.1f us (%.2f TB/s over 4 tensors)   bwd %.1f us (%.2f TB/s over 6 tensors)" % (tf, 4 * mb / tf, tb, 6 * mb / tb))

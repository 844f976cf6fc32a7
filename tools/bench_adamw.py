"""FusedAdamW (csrc/adamw.hip) on a BLIP-sized parameter set: ms per step and effective HBM rate (30 B / parameter)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bridgeqa_amd import fusion_ops
from bridgeqa_amd.optim import FusedAdamW
fusion_ops.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
shapes = [(768, 768)] * 200 + [(3072, 768)] * 40 + [(768, 3072)] * 40 + [(30524, 768)] * 2 + [(768,)] * 600
params = [torch.nn.Parameter(torch.randn(*s, device="cuda") * 0.02) for s in shapes]
for p in params:
    p.grad = torch.randn_like(p) * 0.01
opt = FusedAdamW(params, lr=5e-4, weight_decay=1e-5)
n = sum(p.numel() for p in params)
for _ in range(3):
    opt.step()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10):
    opt.step()
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 10
print("params %.1f M  %.3f ms/step  %.2f TB/s (28 B/param fp32 streams + 2 B shadow when present)" % (n / 1e6, ms, n * 28 / ms / 1e9))

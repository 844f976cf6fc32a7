"""forward / backward error of the fused attention against an fp32 composition"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bridgeqa_amd import _ext
for B, H, L, amp in ((1, 1, 32, 1.5), (2, 12, 197, 1.5), (1, 4, 1025, 1.5), (1, 4, 1025, 0.5), (1, 4, 1025, 3.0)):
    g = torch.Generator().manual_seed(L)
    qkv = (torch.randn(B, L, 3, H, 64, generator=g) * amp).cuda().to(torch.bfloat16)
    go = torch.randn(B, L, H, 64, generator=g).cuda().to(torch.bfloat16)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    out, lse = _ext.attn_fwd(q, k, v, 0.125)
    qf, kf, vf = (t.double().permute(0, 2, 1, 3).detach().requires_grad_(True) for t in (q, k, v))
    s = torch.matmul(qf, kf.transpose(-1, -2)) * 0.125
    want = torch.matmul(torch.softmax(s, -1), vf).permute(0, 2, 1, 3)
    wl = torch.logsumexp(s, -1) / math.log(2.0)
    want.backward(go.double())
    dqkv = torch.empty_like(qkv)
    _ext.attn_bwd(q, k, v, out, lse, go, 0.125, dqkv[:, :, 0], dqkv[:, :, 1], dqkv[:, :, 2])
    rel = lambda a, b: ((a.double() - b).norm() / b.norm()).item()
    print("B%d H%d L%d amp %.1f: out rel %.2e  lse max abs %.2e  dq %.2e dk %.2e dv %.2e" % (
        B, H, L, amp, rel(out, want), (lse.double() - wl).abs().max().item(),
        rel(dqkv[:, :, 0].permute(0, 2, 1, 3), qf.grad), rel(dqkv[:, :, 1].permute(0, 2, 1, 3), kf.grad),
        rel(dqkv[:, :, 2].permute(0, 2, 1, 3), vf.grad)))

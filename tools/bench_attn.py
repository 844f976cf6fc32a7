"""Micro-benchmark of the fused attention kernels at the ViT shape of config c3 (B=16, H=12, L=1025, D=64)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bridgeqa_amd import _ext, fusion_ops

def timeit(f, n=10):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

for B, H, L in ((16, 12, 1025), (16, 12, 901), (32, 12, 4097)):
    qkv = torch.randn(B, L, 3, H, 64, device="cuda").to(torch.bfloat16).requires_grad_(True)
    go = torch.randn(B, L, H, 64, device="cuda").to(torch.bfloat16)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    if hasattr(_ext, "attn_set_persistent"):   # A/B: one workgroup per block against the resident grid (round 6)
        prev = _ext.attn_set_persistent(0)
        t0 = timeit(lambda: _ext.attn_fwd(q.detach(), k.detach(), v.detach(), 0.125), 30)
        o0, l0 = _ext.attn_fwd(q.detach(), k.detach(), v.detach(), 0.125)
        dq0 = torch.empty_like(qkv)
        tb0 = timeit(lambda: _ext.attn_bwd(q.detach(), k.detach(), v.detach(), o0, l0, go, 0.125, dq0[:, :, 0], dq0[:, :, 1], dq0[:, :, 2]), 30)
        _ext.attn_set_persistent(prev)
        print("B=%d H=%d L=%d  per-block launch: fwd %.3f ms  bwd %.3f ms" % (B, H, L, t0, tb0))
    tf = timeit(lambda: _ext.attn_fwd(q.detach(), k.detach(), v.detach(), 0.125), 30)
    out, lse = _ext.attn_fwd(q.detach(), k.detach(), v.detach(), 0.125)
    dqkv = torch.empty_like(qkv)
    tb = timeit(lambda: _ext.attn_bwd(q.detach(), k.detach(), v.detach(), out, lse, go, 0.125, dqkv[:, :, 0], dqkv[:, :, 1], dqkv[:, :, 2]), 30)
    fl = 4.0 * B * H * L * L * 64
    print("B=%d H=%d L=%d  fwd %.3f ms (%.0f TFLOP/s)  bwd %.3f ms (%.0f TFLOP/s on 2.5x fwd flops; %.0f on the 3.5x actually executed)"
          % (B, H, L, tf, fl / tf / 1e9, tb, 2.5 * fl / tb / 1e9, 3.5 * fl / tb / 1e9))
    if L <= 1100:
        def torch_path():
            qq = qkv.detach().requires_grad_(True)
            c, _ = fusion_ops.attention(qq[:, :, 0], qq[:, :, 1], qq[:, :, 2], None, 0.125)
            c.backward(go)
        print("   torch composition fwd+bwd %.3f ms  vs fused %.3f ms" % (timeit(torch_path, 3), tf + tb))

"""A few launches of the fused attention kernels (ViT shape of config c3, then the c5 shape) for rocprofv3 --pmc passes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bridgeqa_amd import _ext

for B, H, L, n in ((16, 12, 1025, 3), (32, 12, 4097, 1)):
    qkv = torch.randn(B, L, 3, H, 64, device="cuda").to(torch.bfloat16)
    go = torch.randn(B, L, H, 64, device="cuda").to(torch.bfloat16)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    dqkv = torch.empty_like(qkv)
    for mask in ((0, 7) if L == 1025 else (0,)):   # block-by-block launches, then the resident-grid kernels (round 6)
        _ext.attn_set_persistent(mask)
        for _ in range(n):
            out, lse = _ext.attn_fwd(q, k, v, 0.125)
            _ext.attn_bwd(q, k, v, out, lse, go, 0.125, dqkv[:, :, 0], dqkv[:, :, 1], dqkv[:, :, 2])
    _ext.attn_set_persistent(0)
    torch.cuda.synchronize()
print("done")

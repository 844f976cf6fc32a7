"""Forward attention kernel at shapes around the ViT's (B=16, H=12, L=1025): what the ragged 1025th token, the number of
workgroup rounds and the sequence length each cost (HIP events, median of 10)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bridgeqa_amd import _ext  # noqa: E402


def med(f, n=10):
    f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        f()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


for B, H, L in ((16, 12, 1025), (16, 12, 1024), (64, 12, 1025), (64, 12, 1024), (16, 12, 2049), (16, 12, 2048), (8, 12, 1025),
                (32, 12, 4097)):
    qkv = torch.randn(B, L, 3, H, 64, device="cuda").to(torch.bfloat16)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    t = med(lambda: _ext.attn_fwd(q, k, v, 0.125))
    fl = 4.0 * B * H * L * L * 64
    wgs = B * H * -(-L // 128)
    print("B=%d H=%d L=%d  fwd %.1f us  %.0f TFLOP/s  (%d workgroups)" % (B, H, L, t * 1e3, fl / t / 1e9, wgs))
    del qkv

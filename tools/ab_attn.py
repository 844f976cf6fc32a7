"""A/B of the attention launch forms in ONE process, interleaved: `python tools/ab_attn.py [mask ...]` times the forward and
the backward at the ViT shapes under every bq_attn_set_persistent mask given (default: 0 and 7), ROUNDS times each in
alternation after a common warm-up, and prints medians (us)."""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bridgeqa_amd import _ext

masks = [int(a, 0) for a in sys.argv[1:]] or [0, 7]
ROUNDS, N = 7, 40


def timed(f):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(N):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / N * 1e3


for B, H, L in ((16, 12, 1025), (16, 12, 1024), (32, 12, 4097)):
    qkv = torch.randn(B, L, 3, H, 64, device="cuda").to(torch.bfloat16)
    go = torch.randn(B, L, H, 64, device="cuda").to(torch.bfloat16)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    dqkv = torch.empty_like(qkv)
    out, lse = _ext.attn_fwd(q, k, v, 0.125)
    fwd = lambda: _ext.attn_fwd(q, k, v, 0.125, out=out)
    bwd = lambda: _ext.attn_bwd(q, k, v, out, lse, go, 0.125, dqkv[:, :, 0], dqkv[:, :, 1], dqkv[:, :, 2])
    for _ in range(3):
        timed(fwd); timed(bwd)
    res = {m: ([], []) for m in masks}
    for _ in range(ROUNDS):
        for m in masks:
            _ext.attn_set_persistent(m)
            res[m][0].append(timed(fwd))
            res[m][1].append(timed(bwd))
    _ext.attn_set_persistent(7)
    fl = 4.0 * B * H * L * L * 64
    for m in masks:
        tf, tb = statistics.median(res[m][0]), statistics.median(res[m][1])
        print("B=%d H=%d L=%d mask 0x%02x: fwd %.1f us (%.0f TFLOP/s)  bwd %.1f us (%.0f TFLOP/s on 2.5x fwd flops)"
              % (B, H, L, m, tf, fl / tf / 1e6, tb, 2.5 * fl / tb / 1e6))

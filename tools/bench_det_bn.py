"""Standalone timing of the detector's BatchNorm(+ReLU, +max-pool) backward (csrc/bn.hip bq_bn_backward: reduce pass, fold,
dx pass) at the SharedMLP shapes of c3: GB/s of its algorithmic traffic (non-pooled: dy and x read twice, dx written =
5 R C 2 B; pooled: x read twice, dx written).  python tools/bench_det_bn.py"""
import torch

from bridgeqa_amd import _ext

SHAPES = [("SA1.l1", 2097152, 64, 64, 0), ("SA1.l3", 2097152, 128, 64, 1), ("SA2.l1", 524288, 128, 32, 0),
          ("SA2.l3", 524288, 256, 32, 1), ("SA3.l1", 131072, 128, 16, 0), ("SA3.l3", 131072, 256, 16, 1)]
dev = torch.device("cuda:0")
for name, R, C, S, pool in SHAPES:
    x = torch.randn(R, C, device=dev).to(torch.bfloat16)
    dy = torch.randn(R // S if pool else R, C, device=dev).to(torch.bfloat16)
    stats = [torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1, torch.randn(C, device=dev) * 0.1,
             torch.rand(C, device=dev) + 0.5]
    for _ in range(3):
        _ext.bn_relu_bwd(dy, x, stats, S, True, bool(pool))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        _ext.bn_relu_bwd(dy, x, stats, S, True, bool(pool))
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    byts = R * C * 2 * (3 if pool else 5)
    print("%-7s R=%8d C=%3d S=%2d pool=%d: %7.1f us  %5.2f TB/s" % (name, R, C, S, pool, us, byts / us / 1e6))

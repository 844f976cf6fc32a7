"""A few launches of the fused SharedMLP backward (csrc/detbwd.hip) at SA1 / SA2's c3 layer shapes for rocprofv3 passes, with
the reduction kernels it follows.  python tools/det_bwd_once.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bridgeqa_amd import _ext  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
# (R, ldx, N, S, pool, need_dx): SA1 layers 0-2, SA2 layers 0-2 at B = 16
for R, ldx, N, S, pool, need_dx in ((2097152, 136, 64, 64, False, False), (2097152, 64, 64, 64, False, True),
                                   (2097152, 64, 128, 64, True, True), (524288, 136, 128, 32, False, True),
                                   (524288, 128, 128, 32, False, True), (524288, 128, 256, 32, True, True)):
    x = torch.randn(R, ldx, device=dev).to(torch.bfloat16)
    y_raw = torch.randn(R, N, device=dev).to(torch.bfloat16)
    dout = torch.randn(R // S if pool else R, N, device=dev).to(torch.bfloat16)
    arg = torch.randint(0, S, (R // S, N), device=dev, dtype=torch.uint8) if pool else None
    w = (torch.randn(N, ((ldx + 63) // 64) * 64, device=dev) * 0.05).to(torch.bfloat16)
    stats = torch.stack([torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev) * 0.1, torch.randn(N, device=dev) * 0.1,
                         torch.rand(N, device=dev) + 0.5, torch.zeros(N, device=dev)])
    for _ in range(3):
        dgb = _ext.bn_bwd_reduce(dout, y_raw, stats, S, True, pool, arg)
        _ext.sa_bwd_fused(x, y_raw, dout, arg, w, stats, dgb, S, True, pool, need_dx)
    torch.cuda.synchronize()
    del x, y_raw, dout, arg

"""Standalone timing of the detector's weight-gradient launches (dW = dY^T X over millions of point rows, cut contraction
with atomics; csrc/gemm.hip gemm64_kernel, P_XC | Q_XC | OUT_F32): GB/s of the operands read once, per workgroup budget.
python tools/bench_det_wgrad.py [wgs ...]"""
import sys

import torch

from bridgeqa_amd import _ext
from bridgeqa_amd.pytorch_utils import _wgrad_pieces

SHAPES = [("SA1.l1", 2097152, 64, 136), ("SA1.l2", 2097152, 64, 64), ("SA1.l3", 2097152, 128, 64),
          ("SA2.l1", 524288, 128, 136), ("SA2.l2", 524288, 128, 128), ("SA2.l3", 524288, 256, 128),
          ("SA3.l1", 131072, 128, 264), ("SA3.l3", 131072, 256, 128), ("SA4.l1", 65536, 128, 264)]
dev = torch.device("cuda:0")
flags = _ext.GEMM_P_XC | _ext.GEMM_Q_XC | _ext.GEMM_OUT_F32
odd = "odd" in sys.argv   # pieces = 8 k - 1: the kernel falls back to the piece-major order (tiles of a piece on different XCDs)
budgets = [int(a) for a in sys.argv[1:] if a != "odd"] or [384]
for name, R, N, K in SHAPES:
    dy = torch.randn(R, N, device=dev).to(torch.bfloat16)
    x = torch.randn(R, K, device=dev).to(torch.bfloat16)
    out = torch.zeros(N, K, device=dev)
    tiles = ((K + 63) // 64) * (N // 64)
    line = "%-7s R=%8d N=%3d K=%3d tiles=%d |" % (name, R, N, K, tiles)
    for wgs in budgets:
        ks = max(1, min((R + 63) // 64, wgs // tiles))
        ks = ks - ks % 8 if ks >= 8 else ks
        ks = ks - 1 if odd and ks >= 8 else ks
        p = [dict(P=x, Q=dy, out=out, ksplit=ks)]
        for _ in range(3):
            _ext.gemm_grouped(p, flags, _ext.EPI_NONE, 64)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            _ext.gemm_grouped(p, flags, _ext.EPI_NONE, 64)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        line += " wgs=%d: %6.1f us %5.2f TB/s |" % (wgs, us, R * (N + K) * 2 / us / 1e6)
    if _ext.wgrad_rows_ok(K, N):
        for wgs in (256, 128, 512):
            for _ in range(3):
                _ext.wgrad_rows(x, dy, out, wgs)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                _ext.wgrad_rows(x, dy, out, wgs)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 100
            line += " ROWS wgs=%d: %6.1f us %5.2f TB/s |" % (wgs, us, R * (N + K) * 2 / us / 1e6)
    print(line)

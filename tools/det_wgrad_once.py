"""A few launches of the detector's weight-gradient kernels at three c3 SharedMLP shapes for rocprofv3 passes: the cut
contraction of gemm64_kernel (384 workgroups, XCD-aware piece order) and bq_wgrad_rows_bf16.  python tools/det_wgrad_once.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bridgeqa_amd import _ext  # noqa: E402
from bridgeqa_amd.pytorch_utils import _wgrad_pieces  # noqa: E402

dev = torch.device("cuda:0")
flags = _ext.GEMM_P_XC | _ext.GEMM_Q_XC | _ext.GEMM_OUT_F32
for R, N, K in ((2097152, 64, 136), (524288, 128, 136), (524288, 256, 128)):
    dy = torch.randn(R, N, device=dev).to(torch.bfloat16)
    x = torch.randn(R, K, device=dev).to(torch.bfloat16)
    out = torch.zeros(N, K, device=dev)
    tiles = ((K + 63) // 64) * (N // 64)
    for _ in range(3):
        _ext.gemm_grouped([dict(P=x, Q=dy, out=out, ksplit=_wgrad_pieces(R, tiles))], flags, _ext.EPI_NONE, 64)
        _ext.wgrad_rows(x, dy, out)
    torch.cuda.synchronize()

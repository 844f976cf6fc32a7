"""How much of the N = 768 shapes' shortfall is tile quantisation: the same kernels on N = 1024 (4 x 256-wide i tiles:
516 tiles over 512 slots) against N = 768 (3 i tiles: 387 tiles)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bridgeqa_amd import _ext  # noqa: E402
from bench_gemm2 import graph_time, rnd  # noqa: E402

dev = torch.device("cuda:0")
for m in (16400, 16384):
    for n, k in ((768, 768), (1024, 768), (768, 3072), (1024, 3072), (768, 2304), (1024, 2304), (2304, 768), (2048, 768)):
        x, w = rnd(m, k), rnd(n, k, scale=0.05)
        b = torch.randn(n, device=dev)
        dy = rnd(m, n)
        t = graph_time(lambda: _ext.gemm_fwd(x, w, b, tile=128))[0]
        t2 = graph_time(lambda: _ext.gemm_dx(dy, w, tile=128))[0]
        print("m=%d n=%4d k=%4d  fwd %.1f us %.0f TF/s   dx(out %d, contraction %d) %.1f us %.0f TF/s"
              % (m, n, k, t, 2.0 * m * n * k / t / 1e6, k, n, t2, 2.0 * m * n * k / t2 / 1e6), flush=True)

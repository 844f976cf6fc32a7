"""Timing-only ablations of gemm128_kernel (csrc/gemm_mid.hip, template VAR): which part of the K loop the time goes to.
VAR bits: 1 no DMA in the loop, 2 no LDS reads, 4 no barriers, 8 no MFMAs.  One process per variant (the variant is read
from the environment once); results are WRONG by construction, only the time matters.

    python tools/bench_gemm_ablate.py            # spawns itself per variant
"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = [("qkv", 16400, 2304, 768), ("proj", 16400, 768, 768), ("fc2", 16400, 768, 3072), ("fc1", 16400, 3072, 768)]


def child():
    import torch
    from bridgeqa_amd import _ext
    from bench_gemm2 import graph_time, rnd
    dev = torch.device("cuda:0")
    out = []
    for name, m, n, k in SHAPES:
        x, w = rnd(m, k), rnd(n, k, scale=0.05)
        b = torch.randn(n, device=dev)
        t = graph_time(lambda: _ext.gemm_fwd(x, w, b, tile=128))
        out.append("%s %.1f" % (name, t[0]))
    print("VAR=%s  " % os.environ.get("BQ_GEMM_MID_VAR", "0") + "  ".join(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        child()
    else:
        for var in (0, 1, 2, 3, 4, 7, 8, 9, 10, 11, 15):
            env = dict(os.environ, BQ_GEMM_MID_VAR=str(var))
            subprocess.call([sys.executable, os.path.abspath(__file__), "child"], env=env)

"""gemm128_kernel (csrc/gemm_mid.hip) at different persistent grid sizes (BQ_GEMM_MID_SLOTS: 1000000 = one tile per
workgroup, i.e. not persistent), one process per setting."""
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
    print("SLOTS=%s  " % os.environ.get("BQ_GEMM_MID_SLOTS", "default") + "  ".join(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        child()
    else:
        for slots in sys.argv[1:] or ("512", "1000000", "256", "768", "1024"):
            env = dict(os.environ, BQ_GEMM_MID_SLOTS=slots)
            subprocess.call([sys.executable, os.path.abspath(__file__), "child"], env=env)

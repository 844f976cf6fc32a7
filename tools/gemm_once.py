"""A few launches of the MFMA GEMM family (csrc/gemm.hip) at the ViT / twin-encoder shapes of config c3 for
rocprofv3 passes (kernel trace, --pmc).  Random bf16 operands.  python tools/gemm_once.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bridgeqa_amd import _ext  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda:0")
M = 16400
rnd = lambda *s, sc=1.0: (torch.randn(*s, device=dev) * sc).to(torch.bfloat16)
for name, N, K in (("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)):
    x, w, dy, b = rnd(M, K), rnd(N, K, sc=0.05), rnd(M, N), torch.randn(N, device=dev)
    pre = rnd(M, K)
    for _ in range(reps):
        if name == "fc1":
            _ext.gemm_fwd(x, w, b, gelu=True, tile=256)
        else:
            _ext.gemm_fwd(x, w, b, tile=256)
        if name == "fc2":
            _ext.gemm_dx(dy, w, pre_act=pre, tile=256)
        else:
            _ext.gemm_dx(dy, w, tile=256)
    torch.cuda.synchronize()
# the grouped weight-gradient launch of the image backward: 12 blocks x 4 linears
probs = []
for blk in range(12):
    for N, K in ((2304, 768), (768, 768), (3072, 768), (768, 3072)):
        probs.append(dict(P=rnd(M, K), Q=rnd(M, N), out=torch.empty(N, K, device=dev)))
for _ in range(max(1, reps // 2)):
    _ext.gemm_grouped(probs, _ext.GEMM_P_XC | _ext.GEMM_Q_XC | _ext.GEMM_OUT_F32, _ext.EPI_NONE, 256)
torch.cuda.synchronize()
# text-side shapes on the small-tile kernel
for m in (320, 80):
    x, w, b = rnd(m, 768), rnd(3072, 768, sc=0.05), torch.randn(3072, device=dev)
    for _ in range(reps):
        _ext.gemm_fwd(x, w, b, gelu=True)
torch.cuda.synchronize()
print("done")

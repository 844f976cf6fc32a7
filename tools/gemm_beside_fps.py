"""How much of the twin K/V GEMM's in-step slowdown (78 us against 45 alone) is the FPS kernel of the prefetched geometry holding
16 CUs (one 160 KB-LDS workgroup per scene) while the persistent 256 x 128 kernel spreads its tiles over 2 x 256 workgroup slots
STATICALLY?  Times the GEMM alone and beside a running FPS launch, full grid and BQ_GEMM_BACKGROUND's half grid.
python tools/gemm_beside_fps.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bridgeqa_amd import _ext
dev = torch.device("cuda:0")
M, N, K = 16720, 1536, 768
x = (torch.randn(M, K, device=dev)).to(torch.bfloat16)
w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
b = torch.randn(N, device=dev)
pts = (torch.rand(16, 40000, 3, device=dev) * torch.tensor([8.0, 8.0, 3.0], device=dev)).contiguous()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def run(beside, background, n=24):
    torch.cuda.synchronize()
    if beside:
        with torch.cuda.stream(sb):
            _ext.furthest_point_sampling(pts, 2048)
    with torch.cuda.stream(sa):
        if beside:
            torch.cuda._sleep(200000)   # let the FPS workgroups settle on their CUs first
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            _ext.gemm_fwd(x, w, b, tile=128, background=background)
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for _ in range(2):
    run(False, False); run(True, False)
for beside in (False, True):
    for background in (False, True):
        ts = sorted(run(beside, background) for _ in range(5))
        print("beside FPS" if beside else "alone     ", "half grid" if background else "full grid", "%.1f us" % ts[2])

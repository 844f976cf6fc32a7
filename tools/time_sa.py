"""Time one SA1-sized SharedMLP forward+backward in fp32 (MIOpen conv path) and bf16 (batched GEMM path)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bridgeqa_amd import fusion_ops
from bridgeqa_amd.pytorch_utils import SharedMLP
torch.manual_seed(0)
for (cin, chans, M, S) in ((135, [64, 64, 128], 2048, 64), (131, [128, 128, 256], 1024, 32), (259, [128, 128, 256], 512, 16)):
    mlp = SharedMLP([cin] + chans, bn=True).cuda().train()
    for dt, cl in ((torch.float32, False), (torch.float32, True)):
        x = torch.randn(16, cin, M, S, device="cuda", dtype=dt)
        if cl:
            x = x.contiguous(memory_format=torch.channels_last)
            mlp = mlp.to(memory_format=torch.channels_last)
        x.requires_grad_(True)
        prev = fusion_ops.set_compute_dtype(dt)
        def step():
            y = mlp(x).max(dim=3)[0].float()
            y.square().mean().backward()
        for _ in range(2):
            t0 = time.time(); step(); torch.cuda.synchronize(); print("  warm", dt, round(time.time() - t0, 3), flush=True)
        t0 = time.time()
        for _ in range(5): step()
        torch.cuda.synchronize()
        print("cin=%d M=%d S=%d %s channels_last=%s: %.2f ms" % (cin, M, S, dt, cl, (time.time() - t0) / 5 * 1e3), flush=True)
        fusion_ops.set_compute_dtype(prev)

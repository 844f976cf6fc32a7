"""SA1's neighbourhood gather alone: (16 x 40000 points x 132 fp32 features) -> (16 x 2048 x 64) rows of 3 + 132 bf16
(padded to 136): 1.1 GB of gathered reads, 570 MB written."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bridgeqa_amd import _ext  # noqa: E402

torch.manual_seed(0)
B, N, C, M, S = 16, 40000, 132, 2048, 64
pc = torch.rand(B, N, 3 + C, device="cuda")
xyz = pc[..., :3].contiguous()
inds = _ext.furthest_point_sampling(xyz, M)
new_xyz = torch.gather(xyz, 1, inds.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
idx = _ext.ball_query(new_xyz, xyz, 0.2, S)
feats = pc[..., 3:]
run = lambda: _ext.group_concat_pm(xyz, new_xyz, feats, idx, 0.2, True, torch.bfloat16, pad_to=8)
out = run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 100
print("group_concat_pm SA1: %.1f us  (%.2f TB/s of output rows)" % (us, out.numel() * 2 / us / 1e6))

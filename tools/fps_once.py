"""SA1 furthest point sampling at config c2/c3 size, a few launches -- the target of the rocprofv3 --pmc passes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bridgeqa_amd import _ext
g = torch.Generator().manual_seed(42)
x = (torch.rand(16, 40000, 3, generator=g) * torch.tensor([8.0, 8.0, 3.0])).contiguous().cuda()
for _ in range(3):
    _ext.furthest_point_sampling(x, 2048)
torch.cuda.synchronize()

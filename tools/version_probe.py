import torch, sys
sys.path.insert(0, "/root/repo")
from bridgeqa_amd import fusion_ops as ops
ops.set_compute_dtype(torch.bfloat16)
dev = torch.device("cuda")
for cap in (False, True):
    lin = torch.nn.Linear(256, 256).to(dev)
    opt = torch.optim.AdamW(lin.parameters(), lr=1e-1, fused=True, capturable=cap)
    x = torch.randn(8, 256, device=dev).to(torch.bfloat16)
    y0 = ops.linear(x, lin.weight, lin.bias).float()
    v0 = lin.weight._version
    y0.square().mean().backward()
    opt.step()
    v1 = lin.weight._version
    y1 = ops.linear(x, lin.weight, lin.bias).float()
    print("capturable", cap, "version", v0, "->", v1, "output changed:", not torch.equal(y0, y1))

"""Do the gradients of one eager step depend on what freed device memory holds?  (a kernel that reads uninitialised memory)"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from bridgeqa_amd import fusion_ops as ops, graphed
import bench
import test_graphed_gpu as T
dev = torch.device("cuda:0")
ops.set_compute_dtype(torch.bfloat16)
def run(fill):
    model, batch = T._setup(dev)
    out = None
    for it in range(2):
        for p in model.parameters():
            p.grad = None
        if it == 1 and fill is not None:
            torch.cuda.synchronize()
            free, total = torch.cuda.mem_get_info()
            n = int(min(free * 0.5, 40e9)) // 4
            x = torch.empty(n, dtype=torch.float32, device=dev)
            if fill == "nan": x.fill_(float("nan"))
            else: x.normal_(0, 1e3)
            torch.cuda.synchronize(); del x     # (freed into the caching allocator: later allocations get these bytes)
        loss = bench.total_loss(model(dict(batch)))
        loss.backward()
    torch.cuda.synchronize()
    return {n: p.grad.detach().float().clone() for n, p in model.named_parameters() if p.grad is not None}, loss.item()
base, l0 = run(None)
for fill in ("rand", "nan"):
    got, l1 = run(fill)
    bad = [(n, ((got[n] - base[n]).norm() / (base[n].norm() + 1e-20)).item()) for n in base if not torch.equal(got[n], base[n])]
    nan = [n for n in got if not torch.isfinite(got[n]).all()]
    print(fill, "loss", l0, l1, "params differing", len(bad), "non-finite", len(nan), sorted(bad, key=lambda t: -t[1])[:5], nan[:5])

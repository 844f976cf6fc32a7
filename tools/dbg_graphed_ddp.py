import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch.distributed as dist
import bench
from bridgeqa_amd import fusion_ops as ops, graphed
from test_graphed_gpu import _setup
dev = torch.device("cuda", 0)
ops.set_compute_dtype(torch.bfloat16)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29671", RANK="0", WORLD_SIZE="1")
dist.init_process_group(backend="nccl", init_method="env://", rank=0, world_size=1)
model, batch = _setup(dev)
def run(force):
    graphed.enable(model)
    model._graphed.force_comm = force
    for _ in range(2):
        bench.total_loss(model(dict(batch))).backward()
    torch.cuda.synchronize()
    g = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    r = model._graphed.reducers
    graphed.disable(model)
    return g, r
a, _ = run(False)
b, _ = run(False)
c, red = run(True)
names = {id(p): n for n, p in model.named_parameters()}
grp = {names[id(p)]: k for k, r in (red or {}).items() for p in r.params}
def cmp(x, y, tag):
    errs = sorted((((x[n] - y[n]).norm() / (x[n].norm() + 1e-12)).item(), n) for n in x)
    print(tag, [(round(e, 4), n, grp.get(n)) for e, n in errs[-6:]])
cmp(a, b, "second enable, no exchange:")
cmp(a, c, "forced exchange:")
dist.destroy_process_group()

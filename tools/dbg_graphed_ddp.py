import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import bench
from bridgeqa_amd import fusion_ops as ops, graphed
from test_graphed_gpu import _setup
dev = torch.device("cuda", 0)
ops.set_compute_dtype(torch.bfloat16)
model, batch = _setup(dev)
def grads():
    torch.cuda.synchronize()
    return {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
def cmp(x, y, tag):
    errs = sorted((((x[n] - y[n]).norm() / (x[n].norm() + 1e-12)).item(), n) for n in x)
    print(tag, [(round(e, 4), n) for e, n in errs[-4:]], flush=True)
def eager():
    for p in model.parameters(): p.grad = None
    l = bench.total_loss(model(dict(batch))); l.backward(); return l.item()
l0 = eager(); e0 = grads(); l1 = eager(); e1 = grads()
print("eager losses", l0, l1); cmp(e0, e1, "eager vs eager:")
graphed.enable(model)
ls = []
gs = []
for it in range(4):
    l = bench.total_loss(model(dict(batch))); l.backward(); ls.append(l.item()); gs.append(grads())
print("runner1 losses", ls)
cmp(e0, gs[1], "eager vs r1 it2:"); cmp(gs[1], gs[2], "r1 it2 vs it3:"); cmp(gs[2], gs[3], "r1 it3 vs it4:")
graphed.disable(model)
graphed.enable(model)
ls2, gs2 = [], []
for it in range(3):
    l = bench.total_loss(model(dict(batch))); l.backward(); ls2.append(l.item()); gs2.append(grads())
print("runner2 losses", ls2)
cmp(gs[1], gs2[1], "r1 it2 vs r2 it2:"); cmp(e0, gs2[1], "eager vs r2 it2:")
l2 = eager(); e2 = grads()
print("eager again loss", l2); cmp(e0, e2, "eager first vs eager after the runners:")

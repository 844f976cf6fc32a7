"""two eager executions of the same c2-sized detector step: are the gradients bitwise identical?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from bridgeqa_amd import fusion_ops as ops, _ext
dev = torch.device("cuda", 0)
ops.set_compute_dtype(torch.bfloat16)
class A: points, cin, image, batch = 40000, 132, 512, 16
torch.manual_seed(0)
model = bench.build_model("c2", A.cin, A.image).to(dev).train()
batch = bench.make_batch(A, "c2", 16, 42, dev)
bufs = {k: v.clone() for k, v in model.named_buffers()}
def run():
    with torch.no_grad():
        for k, v in model.named_buffers(): v.copy_(bufs[k])   # (the running statistics are the centre the stored pre-activations are rounded around)
    for p in model.parameters(): p.grad = None
    l = bench.det_loss(model(dict(batch))); l.backward(); torch.cuda.synchronize()
    return l.item(), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
for det in (True, False):
    _ext.DETERMINISTIC_SCATTER[0] = det
    run()
    (l1, g1), (l2, g2) = run(), run()
    diff = sorted((((g1[n] - g2[n]).norm() / (g1[n].norm() + 1e-20)).item(), n) for n in g1)
    nz = [d for d in diff if d[0] > 0]
    print("deterministic scatter" if det else "atomic scatter", "losses", l1, l2, "| parameters with different gradients: %d of %d" % (len(nz), len(diff)),
          "| worst:", [(round(e, 6), n) for e, n in diff[-3:]], flush=True)
import time
for det in (True, False):
    _ext.DETERMINISTIC_SCATTER[0] = det
    for _ in range(2): run()
    t0 = time.perf_counter()
    for _ in range(5): run()
    print("eager c2 step ms:", det, (time.perf_counter() - t0) / 5 * 1e3)

import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from bridgeqa_amd import fusion_ops as ops, loss_helper as lh, _ext
dev = torch.device("cuda", 0)
ops.set_compute_dtype(torch.bfloat16)
class A: points, cin, image, batch = 40000, 132, 512, 16
torch.manual_seed(0)
model = bench.build_model("c2", A.cin, A.image).to(dev)
batch = bench.make_batch(A, "c2", 16, 42, dev)
dd = model(dict(batch))
print("fused_ok:", type(lh._fused_ok(dd)))
for k in lh._FUSED_FLOAT + lh._FUSED_INT + ("seed_inds",):
    v = dd[k]
    print(k, tuple(v.shape), v.dtype, "contig" if v.is_contiguous() else v.stride(), "base" if v._base is not None else "-", v.requires_grad)
for fused in (True, False, True):
    lh.FUSED_DET_LOSS[0] = fused
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    dd = model(dict(batch))
    torch.cuda.synchronize()
    e0.record()
    loss = bench.det_loss(dd)
    e1.record()
    g = torch.autograd.grad(loss, [dd["center"], dd["vote_xyz"]], retain_graph=True)
    e2.record()
    torch.cuda.synchronize()
    print("fused" if fused else "torch", "loss %.6f fwd %.3f ms bwd(partial) %.3f ms" % (loss.item(), e0.elapsed_time(e1), e1.elapsed_time(e2)))

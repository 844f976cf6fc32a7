"""c5-sized phased step (B=32, N=80000, 1024^2) with switches: python tools/dbg_c5.py [nodet] [nogrid] [nofused] [eagergeo]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from bridgeqa_amd import fusion_ops as ops, _ext, loss_helper
from bridgeqa_amd.optim import FusedAdamW
from bridgeqa_amd.pipeline import PhasedTrainStep
flags = set(sys.argv[1:])
if "nodet" in flags: _ext.DETERMINISTIC_SCATTER[0] = False
if "nogrid" in flags: _ext.BALL_QUERY_GRID_MIN_N[0] = 1 << 30
if "nofused" in flags: loss_helper.FUSED_DET_LOSS[0] = False
dev = torch.device("cuda", 0)
ops.set_compute_dtype(torch.bfloat16)
class A: points, cin, image, batch = 80000, 132, 1024, 32
torch.manual_seed(0)
model = bench.build_model("c3", A.cin, A.image).to(dev).train()
opt = FusedAdamW(model.parameters(), lr=1e-4, weight_decay=1e-5, grad_clip_value=1.0)
batch = bench.make_batch(A, "c3", 32, 43, dev)
pipe = PhasedTrainStep(model, batch, bench.det_loss, bench.fusion_loss, opt, use_graphs=True, next_batch=batch,
                       eager_phases=("geometry",) if "eagergeo" in flags else ())
pipe.capture(warmup=1, keep_warmup_updates=True)
print("captured", flush=True)
for step in range(3):
    loss = pipe.step(); pipe.wait(); torch.cuda.synchronize()
    print("step", step, loss.item(), flush=True)
print("OK", sorted(flags))

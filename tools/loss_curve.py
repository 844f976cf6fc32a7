"""End-to-end sanity of the captured training step: replay the phased graphs on ONE fixed synthetic batch and print the
loss -- it must go down (the optimizer's updates must actually reach the operands the next forward uses)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bridgeqa_amd import fusion_ops
fusion_ops.set_compute_dtype(torch.bfloat16)
import bench
from bridgeqa_amd.pipeline import PhasedTrainStep

dev = torch.device("cuda")
sys.argv = ["bench.py"]
args = bench.parse()
torch.manual_seed(0)
model = bench.build_model("c3", args.cin, args.image).to(dev)
batch = bench.make_batch(args, "c3", args.batch, 42, dev)
if os.environ.get("BQ_TORCH_ADAMW") == "1":
    opt = torch.optim.AdamW(model.parameters(), lr=float(os.environ.get("LR", "1e-4")), weight_decay=1e-5, fused=True, capturable=True)
else:
    from bridgeqa_amd.optim import FusedAdamW
    opt = FusedAdamW(model.parameters(), lr=float(os.environ.get("LR", "1e-4")), weight_decay=1e-5)
pipe = PhasedTrainStep(model, batch, bench.det_loss, bench.fusion_loss, opt, use_graphs=True).capture(warmup=3)
vals = []
for i in range(int(os.environ.get("STEPS", "60"))):
    l = pipe.step()
    if i % 5 == 0:
        pipe.wait(); torch.cuda.synchronize(); vals.append(round(l.item(), 4))
print("loss every 5 steps:", vals)
print("monotone-ish decrease:", vals[-1] < 0.7 * vals[0])

"""Which Python call sites launch the small ATen kernels of the c3 step?  Eager step under torch.profiler with
stacks; kernels are attributed to the innermost bridgeqa_amd frame (forward) or autograd node (backward)."""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from bridgeqa_amd import fusion_ops
fusion_ops.set_compute_dtype(torch.bfloat16)
import bench

dev = torch.device("cuda")
sys.argv = ["bench.py"]
args = bench.parse()
args.cin = 132
torch.manual_seed(0)
WL = os.environ.get("WHO_WORKLOAD", "c3")   # c2: the detector stage alone
model = bench.build_model(WL, 132, args.image).to(dev)
model.train()
batch = bench.make_batch(args, WL, 16, 42, dev)
params = [p for p in model.parameters()]


def step():
    for p in params:
        p.grad = None
    fusion_ops.new_step(dev)
    prev = fusion_ops.set_overlap(False)   # single-stream fusion, as the phased step runs it
    try:
        loss = bench.total_loss(model(dict(batch)))
        fusion_ops.begin_deferred_wgrad()  # weight gradients parked and flushed once, as in pipeline.PhasedTrainStep
        try:
            loss.backward()
        finally:
            fusion_ops.flush_deferred_wgrad()
    finally:
        fusion_ops.set_overlap(prev)
    return loss


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()

pat = sys.argv[1] if len(sys.argv) > 1 else "elementwise"
evs = prof.events()
agg = collections.defaultdict(lambda: [0, 0.0])
for e in evs:
    if e.device_type.name != "CPU":
        continue
    kern = [k for k in e.kernels] if hasattr(e, "kernels") else []
    if not kern:
        continue
    for k in kern:
        kn = k.name
        if not any(p in kn for p in ("at::native", "rocclr", "Cijk")):
            continue
        short = kn.split("<")[0][-40:] + "|" + (kn.split("at::native::")[2][:40] if kn.count("at::native::") > 1 else "")
        frames = [f for f in (e.stack or []) if "bridgeqa_amd" in f or "bench.py" in f]
        site = frames[0].split("/")[-1][:70] if frames else "(autograd) "
        # walk up to the enclosing autograd node name for backward kernels
        par = e.cpu_parent
        node = ""
        while par is not None:
            if "Backward" in par.name or "autograd::engine" in par.name:
                node = par.name
                break
            par = par.cpu_parent
        key = (short, e.name, str(e.input_shapes)[:60], site if not node else node[:60])
        agg[key][0] += 1
        agg[key][1] += k.duration
TOP = int(sys.argv[2]) if len(sys.argv) > 2 else 70
SKIP = sys.argv[3].split(",") if len(sys.argv) > 3 else []   # drop call sites containing any of these (e.g. loss_helper,pointnet2)
items = [(k, v) for k, v in agg.items() if not any(x in k[3] for x in SKIP)]
print("%d launches, %.1f us in the listed classes" % (sum(v[0] for _, v in items), sum(v[1] for _, v in items)))
for k, v in sorted(items, key=lambda x: -x[1][1])[:TOP]:
    print("%4d %8.1fus  %-60s %-28s %-60s %s" % (v[0], v[1], k[0], k[1][:28], k[2], k[3]))

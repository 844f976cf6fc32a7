"""Which module / autograd node launches the library (ATen / rocclr / Cijk) kernels of the c3 step?  One eager step under
torch.profiler; every module's forward runs inside a record_function scope named after the module path, so a launch is
attributed to the innermost scope around it (forward) or to the autograd node that issued it (backward).

    python tools/who_launches.py [pattern] [top] [skip,substrings]      WHO_WORKLOAD=c2: the detector stage alone
"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity, record_function
from bridgeqa_amd import fusion_ops
fusion_ops.set_compute_dtype(torch.bfloat16)
import bench

dev = torch.device("cuda")
sys.argv, argv = ["bench.py"], sys.argv
args = bench.parse()
args.cin = 132
torch.manual_seed(0)
WL = os.environ.get("WHO_WORKLOAD", "c3")
model = bench.build_model(WL, 132, args.image).to(dev)
model.train()
batch = bench.make_batch(args, WL, 16, 42, dev)
params = [p for p in model.parameters()]


def scoped(mod, name):
    orig = mod.forward

    def fwd(*a, **k):
        with record_function("mod:" + name):
            return orig(*a, **k)
    mod.forward = fwd


for name, mod in model.named_modules():
    if name and name.count(".") <= 6:
        scoped(mod, name)


def step():
    for p in params:
        p.grad = None
    fusion_ops.new_step(dev)
    prev = fusion_ops.set_overlap(False)   # single-stream fusion, as the phased step runs it
    try:
        with record_function("mod:LOSS"):
            pass
        dd = model(dict(batch))
        with record_function("mod:LOSS"):
            loss = bench.total_loss(dd)
        fusion_ops.begin_deferred_wgrad()
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()

pat = argv[1] if len(argv) > 1 else ""
TOP = int(argv[2]) if len(argv) > 2 else 120
SKIP = argv[3].split(",") if len(argv) > 3 else []
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.device_type.name != "CPU" or not getattr(e, "kernels", None):
        continue
    for k in e.kernels:
        kn = k.name
        if not any(p in kn for p in ("at::native", "rocclr", "Cijk")) or pat not in kn:
            continue
        par, scope, node = e.cpu_parent, None, None
        while par is not None:
            if scope is None and par.name.startswith("mod:"):
                scope = par.name[4:]
            if node is None and ("Backward" in par.name or "autograd::engine" in par.name):
                node = par.name.replace("autograd::engine::evaluate_function: ", "")
            par = par.cpu_parent
        where = node if node is not None else (scope or "(top level)")
        key = (where[:70], e.name[:28], str(e.input_shapes)[:50])
        agg[key][0] += 1
        agg[key][1] += k.duration
items = [(k, v) for k, v in agg.items() if not any(x in k[0] for x in SKIP)]
print("%d launches, %.1f us in the listed classes" % (sum(v[0] for _, v in items), sum(v[1] for _, v in items)))
by_where = collections.defaultdict(lambda: [0, 0.0])
for (w, _, _), v in items:
    by_where[w][0] += v[0]; by_where[w][1] += v[1]
print("---- by owner")
for w, v in sorted(by_where.items(), key=lambda x: -x[1][0])[:TOP]:
    print("%4d %8.1fus  %s" % (v[0], v[1], w))
print("---- by owner / op / shapes")
for k, v in sorted(items, key=lambda x: -x[1][0])[:TOP]:
    print("%4d %8.1fus  %-70s %-28s %s" % (v[0], v[1], k[0], k[1], k[2]))

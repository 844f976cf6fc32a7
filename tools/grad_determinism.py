"""Is one backward bit-reproducible?  The same forward + backward (eager, one batch, parameters untouched) N times; per parameter
the number of executions whose gradient differs in any bit from the first one's, and the largest difference.

    python tools/grad_determinism.py [c2|c3] [N]

Why it exists (round 6): the four-step loop test found a second loss trajectory in one execution out of seven, whatever the mode
-- some fp32 sum is not order-stable; this names the parameters whose gradients move.  c2 (the detector stage), 24 executions,
first run: one loss value; 66 of 77 parameter gradients bit-identical every time (every SharedMLP of the SA modules: sa_bwd /
wgrad_rows, the gather gradients); 11 moved by 2e-7 .. 6e-7 of their largest element -- the convolutions of the FP modules, of
the voting module and of the proposal head, whose weight gradients were cut contractions summed with fp32 atomics.  Since
pytorch_utils._cut_dw (the pieces as separate problems of one launch, summed in a fixed order): 0 of 77 over 16 executions.
(c3 is not comparable this way: fusion_ops.new_step draws new dropout masks.)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bridgeqa_amd import fusion_ops  # noqa: E402

fusion_ops.set_compute_dtype(torch.bfloat16)
import bench  # noqa: E402

WL = sys.argv[1] if len(sys.argv) > 1 else "c2"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dev = torch.device("cuda")
sys.argv = ["bench.py"]
args = bench.parse()
args.cin = 132
torch.manual_seed(0)
model = bench.build_model(WL, 132, args.image).to(dev)
model.train()
batch = bench.make_batch(args, WL, 16, 42, dev)
names = [n for n, _ in model.named_parameters()]
params = [p for _, p in model.named_parameters()]


# (the buffers are part of the state: SharedMLP pre-activations are stored relative to BatchNorm's running mean --
# pytorch_utils.CENTER_PREACT -- so a forward depends on it at bf16 rounding level; every execution starts from the same ones)
BUFS = {n: b.clone() for n, b in model.named_buffers()}


def grads():
    for n, b in model.named_buffers():
        b.copy_(BUFS[n])
    for p in params:
        p.grad = None
    fusion_ops.new_step(dev)
    dd = model(dict(batch))
    loss = bench.total_loss(dd) if WL != "c2" else bench.det_loss(dd)
    fusion_ops.begin_deferred_wgrad()
    try:
        loss.backward()
    finally:
        fusion_ops.flush_deferred_wgrad()
    torch.cuda.synchronize()
    return float(loss.detach()), [None if p.grad is None else p.grad.detach().clone() for p in params]


l0, g0 = grads()
moved = {}
losses = {l0}
for it in range(1, N):
    l, g = grads()
    losses.add(l)
    for n, a, b in zip(names, g0, g):
        if a is None or b is None:
            continue
        if not torch.equal(a, b):
            d = (a.float() - b.float()).abs().max().item() / max(a.float().abs().max().item(), 1e-30)
            c, m = moved.get(n, (0, 0.0))
            moved[n] = (c + 1, max(m, d))
print("%s: %d executions, %d distinct loss values %s" % (WL, N, len(losses), sorted(losses)[:4]))
print("%d of %d parameters had a gradient that differed from the first execution's at least once" % (len(moved), len(names)))
for n, (c, m) in sorted(moved.items(), key=lambda x: -x[1][1])[:40]:
    print("  %-70s differed in %2d of %d   max |diff| / max |grad| %.2e" % (n, c, N - 1, m))

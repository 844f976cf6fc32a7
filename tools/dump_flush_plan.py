"""Which weight-gradient problems does each deferred flush of a c3 step hold, and how does fusion_wgrad plan them?
One eager PhasedTrainStep with flush_deferred_items wrapped: per flush the (rows, N, K) histogram, the planner's groups
and the HIP-event time of the flush."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from bridgeqa_amd import fusion_ops, fusion_wgrad  # noqa: E402
from bridgeqa_amd.optim import FusedAdamW  # noqa: E402
from bridgeqa_amd.pipeline import PhasedTrainStep  # noqa: E402

sys.argv = ["bench.py"]
args = bench.parse()
dev = torch.device("cuda:0")
fusion_ops.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
model = bench.build_model("c3", args.cin, args.image).to(dev)
batch = bench.make_batch(args, "c3", args.batch, 42, dev)
opt = FusedAdamW(model.parameters(), lr=5e-4, weight_decay=1e-5, grad_clip_value=1.0)
pipe = PhasedTrainStep(model, batch, bench.det_loss, bench.fusion_loss, opt, use_graphs=False, next_batch=batch)
pipe.capture(warmup=2)
orig = fusion_wgrad.flush_deferred_items
log = []


def wrapped(items):
    hist = collections.Counter()
    for it in items:
        g, x = it[0], it[1]
        rows = fusion_wgrad._nrows(g)
        hist[(rows, g.shape[-1], x.shape[-1], len(it) > 4 and it[4] is not None)] += 1
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    orig(items)
    e1.record()
    log.append((hist, e0, e1))


fusion_wgrad.flush_deferred_items = wrapped
fusion_ops.flush_deferred_items = wrapped
import bridgeqa_amd.fusion_wgrad as fw  # noqa: E402
fw_flush = fw.flush_deferred_wgrad
fw.flush_deferred_wgrad = lambda: wrapped(fw.take_deferred_wgrad())
fusion_ops.flush_deferred_wgrad = fw.flush_deferred_wgrad
pipe.eager_step()
torch.cuda.synchronize()
for hist, e0, e1 in log:
    print("flush: %d records, %.2f ms (eager: host time included)" % (sum(hist.values()), e0.elapsed_time(e1)))
    for (rows, n, k, second), c in sorted(hist.items(), key=lambda kv: -kv[0][0]):
        print("   %3d x  rows %6d  N %5d  K %5d %s" % (c, rows, n, k, "(second row source)" if second else ""))

"""Round 6: which switch makes the eager loop and the graph-replayed loop disagree on a gradient?  One forward + backward of the
test model (tests/test_graphed_gpu.py) per arm -- the hoisted decoder K/V input gradients chained (seq) or as one grouped launch +
sum (grp), eager or graphed -- and the rel-L2 of chosen parameters' gradients between arms, then every parameter that differs
between grp_graphed and grp_eager: the detector's gradients differ by 1e-7 at FP2 and by 8e-3 at SA1."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from bridgeqa_amd import fusion_ops as ops
import test_graphed_gpu as T
dev = torch.device("cuda:0")
ops.set_compute_dtype(torch.bfloat16)
res = {}
for name, flag, mode in (("seq_eager", (True, False), "eager"), ("grp_eager", (True, True), "eager"), ("grp_graphed", (True, True), "graphed"),
                         ("grp_graphed2", (True, True), "graphed"), ("seq_graphed", (True, False), "graphed")):
    ops._HOIST_GROUPED[0], ops._HOIST_GROUPED_DX[0] = flag
    res[name] = T._grads_once(dev, mode)[0]
names = list(res["seq_eager"])
def rel(a, b): return ((a - b).norm() / (b.norm() + 1e-20)).item()
import re
groups = {"vit": "visual_encoder.blocks.5.attn.qkv.weight", "twin": "text_encoder.encoder.layer.3.attention.self.query.weight",
          "dec_cross_k": "text_decoder.bert.encoder.layer.4.crossattention.self.key.weight", "det": "detection_backbone.sa2.mlp_module.layer1.conv.weight",
          "objlin": "object_feat_linear.0.weight"}
for a, b in (("grp_eager", "seq_eager"), ("grp_graphed", "seq_eager"), ("grp_graphed", "grp_eager"), ("grp_graphed2", "grp_graphed"), ("seq_graphed", "seq_eager")):
    out = []
    for g, pat in groups.items():
        n = next(x for x in names if pat in x)
        out.append("%s %.2e" % (g, rel(res[a][n], res[b][n])))
    tot = torch.cat([res[a][n].flatten() for n in names]); tob = torch.cat([res[b][n].flatten() for n in names])
    print(a, "vs", b, " ".join(out), "ALL %.2e" % rel(tot, tob))
print("---- grp_graphed vs grp_eager, every differing parameter")
for n in names:
    d = rel(res["grp_graphed"][n], res["grp_eager"][n])
    if d > 0:
        print("%.3e  %s  %s" % (d, n, tuple(res["grp_eager"][n].shape)))

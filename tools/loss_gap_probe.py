"""Where does the bf16 detector path's convergence gap at c2 come from?  The fp32 torch composition of
tools/loss_curve.py with ONE class of bf16 rounding injected at a time (value rounded, gradient passed straight through
unless the variant says otherwise), 200 steps on the fixed c2 batch, next to the plain fp32 run and the bf16 kernel path:

  xyz      the three normalised-offset channels of every grouped tensor (QueryAndGroup's output) rounded to bf16
  grouped  the whole grouped tensor rounded (what csrc/pn2_ops.hip group_concat_pm writes)
  acts     every SharedMLP layer's output (after BatchNorm + ReLU) rounded
  pre      every SharedMLP layer's convolution output (before BatchNorm) rounded
  grads    the gradient arriving at every SharedMLP layer's output rounded
  weights  every SharedMLP convolution weight rounded in the forward (fp32 master weights, as the shadows)

    python tools/loss_gap_probe.py [--steps 200] [--reps 2] [--variants fp32,bf16,xyz,...] [--out file.json]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import loss_curve  # noqa: E402
from bridgeqa_amd import pointnet2_utils, pytorch_utils  # noqa: E402

MODE = [None]


def _round(x):
    return x + (x.to(torch.bfloat16).to(x.dtype) - x).detach()


_orig_group = pointnet2_utils.QueryAndGroup.forward
_orig_conv = pytorch_utils.Conv2d.forward


def _group(self, *a, **kw):
    out = _orig_group(self, *a, **kw)
    m = MODE[0] or ()
    if "xyz" not in m and "grouped" not in m:
        return out
    first, rest = (out[0], out[1:]) if isinstance(out, tuple) else (out, None)
    if first.dtype == torch.float32:
        first = _round(first) if "grouped" in m else torch.cat([_round(first[:, :3]), first[:, 3:]], dim=1)
    return first if rest is None else (first,) + tuple(rest)


def _conv(self, x):
    m = MODE[0] or ()
    if not any(k in m for k in ("acts", "pre", "grads", "weights")) or x.dtype != torch.float32:
        return _orig_conv(self, x)
    # the layer by hand (conv -> bn -> activation, children as named in pytorch_utils.Conv2d), every requested class applied
    conv = self.conv
    w = _round(conv.weight) if "weights" in m else conv.weight
    y = F.conv2d(x, w, conv.bias)
    if "pre" in m:
        y = _round(y)
    for name, mod in self.named_children():
        if name != "conv":
            y = mod(y)
    if "acts" in m:
        y = _round(y)
    if "grads" in m and y.requires_grad:
        y.register_hook(lambda g: g.to(torch.bfloat16).to(g.dtype))
    return y


pointnet2_utils.QueryAndGroup.forward = _group
pytorch_utils.Conv2d.forward = _conv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--variants", default="fp32,bf16,xyz,grouped,acts,pre,grads,weights")
    ap.add_argument("--out", default=None)
    ap.add_argument("--pair", default=None, metavar="r06_loss_curve.json",
                    help="round 6: one run of every variant per (weights, batch) seed of that file, paired with its fp32 finals; "
                         "variants may be sums of classes (pre+acts+grouped+grads+weights)")
    ap.add_argument("--seeds", type=int, default=12)
    args = ap.parse_args()
    tail = 10
    if args.pair:
        import math
        prev = json.load(open(args.pair))["runs"][:args.seeds]
        out = {}
        for v in args.variants.split(","):
            MODE[0] = tuple(v.split("+"))
            rel = []
            for r in prev:
                c = loss_curve.run_curve("c2", torch.float32, args.steps, seed=r["seed"], batch_seed=42 + r["seed"])
                f32 = 0.5 * (r["fp32"]["final"] + r["fp32_repeat"]["final"])
                rel.append((sum(c[-tail:]) / tail - f32) / f32)
            m = sum(rel) / len(rel)
            sd = math.sqrt(sum((x - m) ** 2 for x in rel) / (len(rel) - 1))
            out[v] = {"paired_rel_diff_vs_fp32_pct": round(100 * m, 2), "standard_error": round(100 * sd / math.sqrt(len(rel)), 2),
                      "per_seed_pct": [round(100 * x, 2) for x in rel]}
            print("%-34s %+6.2f %% +- %.2f (SE, %d seeds)" % (v, 100 * m, 100 * sd / math.sqrt(len(rel)), len(rel)), flush=True)
        MODE[0] = None
        if args.out:
            json.dump({"what": "fp32 composition with classes of bf16 rounding injected, paired with the fp32 finals of " + args.pair +
                               " (tools/loss_gap_probe.py --pair); the kernel path itself: see that file",
                       "steps": args.steps, "results": out}, open(args.out, "w"), indent=1)
        return
    res = {}
    for rep in range(args.reps):
        for v in args.variants.split(","):
            MODE[0] = tuple(v.split("+")) if v not in ("fp32", "bf16") else None
            c = loss_curve.run_curve("c2", torch.bfloat16 if v == "bf16" else torch.float32, args.steps)
            fin = sum(c[-tail:]) / tail
            res.setdefault(v, []).append({"first": c[0], "final": fin, "drop": fin / c[0]})
            print("rep %d %-8s first %.4f final %.4f  loss drop x%.3f" % (rep, v, c[0], fin, fin / c[0]), flush=True)
    base = sum(r["final"] for r in res["fp32"]) / len(res["fp32"]) if "fp32" in res else None
    for v, rs in res.items():
        f = sum(r["final"] for r in rs) / len(rs)
        print("%-8s mean final %.4f  drop x%.3f%s" % (v, f, sum(r["drop"] for r in rs) / len(rs),
                                                      "" if base is None else "  vs fp32 %+.1f %%" % (100 * (f / base - 1))))
    if args.out:
        json.dump({"what": __doc__.split("\n\n")[0], "steps": args.steps, "results": res}, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()

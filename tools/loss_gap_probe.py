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
    m = MODE[0]
    if m not in ("xyz", "grouped"):
        return out
    first, rest = (out[0], out[1:]) if isinstance(out, tuple) else (out, None)
    if first.dtype == torch.float32:
        first = _round(first) if m == "grouped" else torch.cat([_round(first[:, :3]), first[:, 3:]], dim=1)
    return first if rest is None else (first,) + tuple(rest)


def _conv(self, x):
    m = MODE[0]
    if m not in ("acts", "pre", "grads", "weights") or x.dtype != torch.float32:
        return _orig_conv(self, x)
    if m == "acts":
        return _round(_orig_conv(self, x))
    if m == "grads":
        y = _orig_conv(self, x)
        if y.requires_grad:
            y.register_hook(lambda g: g.to(torch.bfloat16).to(g.dtype))
        return y
    # pre / weights: the layer by hand (conv -> bn -> activation, children as named in pytorch_utils.Conv2d)
    conv = self.conv
    w = _round(conv.weight) if m == "weights" else conv.weight
    y = F.conv2d(x, w, conv.bias)
    if m == "pre":
        y = _round(y)
    for name, mod in self.named_children():
        if name != "conv":
            y = mod(y)
    return y


pointnet2_utils.QueryAndGroup.forward = _group
pytorch_utils.Conv2d.forward = _conv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--variants", default="fp32,bf16,xyz,grouped,acts,pre,grads,weights")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    tail = 10
    res = {}
    for rep in range(args.reps):
        for v in args.variants.split(","):
            MODE[0] = v if v not in ("fp32", "bf16") else None
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

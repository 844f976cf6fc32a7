"""Convergence evidence for the bf16 kernel path: the SAME training run -- same initial weights, same
fixed synthetic batch, same optimizer arithmetic (optim.FusedAdamW, lr 1e-4, the reference's clip_grad_value_(1.0)), every
stochastic layer switched off (dropout / drop-path p = 0, so that the two runs see the same function) -- once on the bf16
HIP kernel path and once on the fp32 torch composition, loss recorded at every step.

    python tools/loss_curve.py [--steps 200] [--out profiles/r03_loss_curve.json]

Workloads: "c2" = the DET-stage hot path at full size (B=16 x 40000 points, C_in=132); "c3s" = the VQA-stage hot path at a
reduced size (B=4 x 8192 points, C_in=132, one 256^2 view: 257 image tokens) -- the fp32 composition of full-size c3 runs
~0.5 s per step.  tests/test_convergence_gpu.py runs a shorter version of the same comparison and asserts the gap.
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

WORKLOADS = {"c2": dict(workload="c2", batch=16, points=40000, image=512),
             "c3s": dict(workload="c3", batch=4, points=8192, image=256)}


class _Args(object):
    cin = 132

    def __init__(self, points, image):
        self.points, self.image = points, image


def run_curve(name, dtype, steps, lr=1e-4, seed=0, round_inputs=False, batch_seed=42):
    """losses (python floats, one per step) of `steps` eager training steps of workload `name` in compute dtype `dtype`;
    round_inputs: the per-point features and the image rounded to bf16 ONCE (everything else as `dtype` says) -- the
    control run: an fp32 run under a perturbation of the size of a single bf16 rounding"""
    import bench
    from bridgeqa_amd import fusion_ops as ops
    from bridgeqa_amd.optim import FusedAdamW
    w = WORKLOADS[name]
    dev = torch.device("cuda:0")
    prev = ops.set_compute_dtype(dtype)
    try:
        torch.manual_seed(seed)
        model = bench.build_model(w["workload"], _Args.cin, w["image"]).to(dev)
        model.train()
        for mod in model.modules():      # the same deterministic function in both runs
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
            if hasattr(mod, "drop_prob"):
                mod.drop_prob = 0.0
        batch = bench.make_batch(_Args(w["points"], w["image"]), w["workload"], w["batch"], batch_seed, dev)
        if round_inputs:
            batch["point_clouds"][..., 3:] = batch["point_clouds"][..., 3:].to(torch.bfloat16).float()
            if "images" in batch:
                batch["images"] = batch["images"].to(torch.bfloat16).float()
        opt = FusedAdamW(model.parameters(), lr=lr, weight_decay=1e-5, grad_clip_value=1.0)
        losses = []
        for _ in range(steps):
            opt.zero_grad(set_to_none=True)
            loss = bench.total_loss(model(dict(batch)))
            loss.backward()
            opt.step()
            losses.append(loss.detach())
        torch.cuda.synchronize()
        return [float(x) for x in torch.stack(losses).cpu()]
    finally:
        ops.set_compute_dtype(prev)


def compare(name, steps, tail=10):
    """fp32 | bf16 kernel path | control (fp32 with the inputs rounded to bf16 once): the control says how far two runs of
    this loss drift apart under a perturbation of bf16 size -- the loss is not a smooth function of the features (vote
    clustering by FPS over PREDICTED votes, nearest-centre objectness labels, max-pool winners)"""
    a = run_curve(name, torch.float32, steps)
    b = run_curve(name, torch.bfloat16, steps)
    c = run_curve(name, torch.float32, steps, round_inputs=True)
    fa, fb, fc = sum(a[-tail:]) / tail, sum(b[-tail:]) / tail, sum(c[-tail:]) / tail
    gap = lambda u: max(abs(x - y) / max(abs(x), 1e-12) for x, y in zip(a, u))
    return {"workload": name, "config": WORKLOADS[name], "steps": steps, "lr": 1e-4,
            "first_loss": {"fp32": a[0], "bf16": b[0], "control": c[0]},
            "final_loss_mean_of_last_%d" % tail: {"fp32": fa, "bf16": fb, "control": fc},
            "final_gap_rel": abs(fa - fb) / abs(fa), "control_final_gap_rel": abs(fa - fc) / abs(fa),
            "loss_drop": {"fp32": fa / a[0], "bf16": fb / b[0], "control": fc / c[0]},
            "max_rel_gap_over_the_curve": gap(b), "control_max_rel_gap_over_the_curve": gap(c),
            "fp32": [round(x, 5) for x in a], "bf16": [round(x, 5) for x in b], "control": [round(x, 5) for x in c]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--out", default=None)
    ap.add_argument("--workloads", default="c2,c3s")
    args = ap.parse_args()
    recs = []
    for name in args.workloads.split(","):
        r = compare(name, args.steps)
        recs.append(r)
        f = r["final_loss_mean_of_last_10"]
        print("%s: first %.4f / %.4f / %.4f  final %.4f (fp32) %.4f (bf16) %.4f (control)  final gap bf16 %.2f %% control %.2f %%  "
              "max gap along the curve bf16 %.2f %% control %.2f %%  loss drop x%.3f / x%.3f / x%.3f"
              % (name, r["first_loss"]["fp32"], r["first_loss"]["bf16"], r["first_loss"]["control"], f["fp32"], f["bf16"],
                 f["control"], 100 * r["final_gap_rel"], 100 * r["control_final_gap_rel"],
                 100 * r["max_rel_gap_over_the_curve"], 100 * r["control_max_rel_gap_over_the_curve"],
                 r["loss_drop"]["fp32"], r["loss_drop"]["bf16"], r["loss_drop"]["control"]), flush=True)
    if args.out:
        with open(args.out, "w") as f:
            json.dump({"what": __doc__.split("\n\n")[0], "runs": recs}, f, indent=1)


if __name__ == "__main__":
    main()

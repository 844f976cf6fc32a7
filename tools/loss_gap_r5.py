"""does taking the SharedMLP outputs from the fp32 accumulators (bq_pwconv_bn_apply, _ext.FP32_PREACT) close
the detector's bf16 convergence gap?  The round-3 protocol of tools/loss_curve.py (200 eager steps on one fixed c2 batch,
FusedAdamW lr 1e-4, clip 1.0, stochastic layers off), several executions per arm:
  fp32            the torch composition
  bf16            the kernel path with FP32_PREACT on (this round's default)
  bf16_stored     the kernel path with the output taken from the stored bf16 pre-activation (round 4's path)

    python tools/loss_gap_r5.py [--steps 200] [--reps 3] [--out profiles/r05_loss_curve.json]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from loss_curve import run_curve  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    from bridgeqa_amd import _ext
    tail = 10
    arms = {"fp32": (torch.float32, True), "bf16": (torch.bfloat16, True), "bf16_stored": (torch.bfloat16, False)}
    res = {k: [] for k in arms}
    for rep in range(a.reps):
        for name, (dt, fp32pre) in arms.items():
            _ext.FP32_PREACT[0] = fp32pre
            c = run_curve(a.workload, dt, a.steps)
            res[name].append({"first": c[0], "final": sum(c[-tail:]) / tail, "curve": [round(x, 4) for x in c]})
            print(rep, name, "first %.4f final %.4f" % (c[0], res[name][-1]["final"]), flush=True)
    _ext.FP32_PREACT[0] = True
    mean = lambda k: sum(r["final"] for r in res[k]) / len(res[k])
    f = mean("fp32")
    summary = {k: {"finals": [round(r["final"], 4) for r in res[k]], "mean": round(mean(k), 4),
                   "gap_to_fp32_mean_pct": round(100 * (mean(k) - f) / f, 2)} for k in arms}
    summary["fp32"]["spread_pct"] = round(100 * (max(r["final"] for r in res["fp32"]) - min(r["final"] for r in res["fp32"])) / f, 2)
    print(json.dumps(summary))
    if a.out:
        json.dump({"what": __doc__.split("\n\n")[0], "workload": a.workload, "steps": a.steps, "summary": summary, "runs": res},
                  open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()

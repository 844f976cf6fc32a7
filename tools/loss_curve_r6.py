"""is the bf16 detector's 200-step training loss different from the fp32 composition's, measured so that
the answer can be read off?  Round 5 compared 3 executions per arm on ONE batch and the fp32 arm alone spread 12.9 % between
identical runs (fp32 atomics under a discontinuous detection loss).  Here: N different (initial weights, synthetic batch) seeds,
both arms on every seed (a PAIRED design: the seed-to-seed variation of the task cancels in the difference), the round-3
protocol otherwise (tools/loss_curve.py: eager steps on one fixed batch per seed, FusedAdamW lr 1e-4, clip 1.0, stochastic
layers off, final loss = mean of the last 10 steps).  Reported: per-arm mean +- sd of the final loss, the paired relative
difference (bf16 - fp32) / fp32 as mean +- standard error, and the fp32 arm's run-to-run noise from a repeat of every seed.

    python tools/loss_curve_r6.py [--steps 200] [--seeds 12] [--out profiles/r06_loss_curve.json]
"""
import argparse
import json
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from loss_curve import run_curve  # noqa: E402


def stats(v):
    m = sum(v) / len(v)
    sd = math.sqrt(sum((x - m) ** 2 for x in v) / max(1, len(v) - 1))
    return m, sd


def preact_arm(a):
    from bridgeqa_amd import _ext
    prev = json.load(open(a.preact_arm))
    rows = []
    _ext.FP32_PREACT[0] = True
    try:
        for r in prev["runs"][:a.seeds]:
            s = r["seed"]
            c1 = run_curve(a.workload, torch.bfloat16, a.steps, seed=s, batch_seed=42 + s)
            c2 = run_curve(a.workload, torch.bfloat16, a.steps, seed=s, batch_seed=42 + s)
            f = 0.5 * (sum(c1[-10:]) / 10 + sum(c2[-10:]) / 10)
            f32 = 0.5 * (r["fp32"]["final"] + r["fp32_repeat"]["final"])
            b16 = 0.5 * (r["bf16"]["final"] + r["bf16_repeat"]["final"])
            rows.append({"seed": s, "bf16_fp32_preact": f, "fp32": f32, "bf16": b16})
            print(s, "fp32 %.4f bf16 %.4f bf16+fp32-preact %.4f" % (f32, b16, f), flush=True)
    finally:
        _ext.FP32_PREACT[0] = False
    rel = [(r["bf16_fp32_preact"] - r["fp32"]) / r["fp32"] for r in rows]
    rel_b = [(r["bf16_fp32_preact"] - r["bf16"]) / r["bf16"] for r in rows]
    (m, sd), (mb, sdb) = stats(rel), stats(rel_b)
    n = math.sqrt(len(rows))
    summary = {"seeds": len(rows), "arm": "bf16 kernel path with _ext.FP32_PREACT (SharedMLP outputs from the fp32 accumulators)",
               "paired_rel_diff_vs_fp32_pct": {"mean": round(100 * m, 2), "standard_error": round(100 * sd / n, 2)},
               "paired_rel_diff_vs_bf16_default_pct": {"mean": round(100 * mb, 2), "standard_error": round(100 * sdb / n, 2)}}
    print(json.dumps(summary))
    if a.out:
        json.dump({"what": "tools/loss_curve_r6.py --preact-arm: does round 5's fp32-accumulator SharedMLP output close the gap?",
                   "summary": summary, "runs": rows}, open(a.out, "w"), indent=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--seeds", type=int, default=12)
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--out", default=None)
    ap.add_argument("--preact-arm", default=None, metavar="PREVIOUS.json",
                    help="run ONLY the bf16 path with the SharedMLP outputs taken from the fp32 accumulators (_ext.FP32_PREACT, "
                         "round 5's opt-in), twice per seed, and pair it with the fp32 / bf16 finals of PREVIOUS.json (same seeds)")
    a = ap.parse_args()
    if a.preact_arm:
        return preact_arm(a)
    tail = 10
    final = lambda c: sum(c[-tail:]) / tail
    rows = []
    for s in range(a.seeds):
        r = {"seed": s}
        for arm, dt in (("fp32", torch.float32), ("fp32_repeat", torch.float32), ("bf16", torch.bfloat16), ("bf16_repeat", torch.bfloat16)):
            c = run_curve(a.workload, dt, a.steps, seed=s, batch_seed=42 + s)
            r[arm] = {"first": c[0], "final": final(c)}
        rows.append(r)
        print(s, " ".join("%s %.4f" % (k, r[k]["final"]) for k in ("fp32", "fp32_repeat", "bf16", "bf16_repeat")), flush=True)
    f32 = [0.5 * (r["fp32"]["final"] + r["fp32_repeat"]["final"]) for r in rows]
    b16 = [0.5 * (r["bf16"]["final"] + r["bf16_repeat"]["final"]) for r in rows]
    rel = [(b - f) / f for b, f in zip(b16, f32)]
    rep32 = [abs(r["fp32"]["final"] - r["fp32_repeat"]["final"]) / (0.5 * (r["fp32"]["final"] + r["fp32_repeat"]["final"])) for r in rows]
    rep16 = [abs(r["bf16"]["final"] - r["bf16_repeat"]["final"]) / (0.5 * (r["bf16"]["final"] + r["bf16_repeat"]["final"])) for r in rows]
    m32, s32 = stats(f32)
    m16, s16 = stats(b16)
    mr, sr = stats(rel)
    se = sr / math.sqrt(len(rel))
    summary = {"seeds": a.seeds, "steps": a.steps, "workload": a.workload,
               "final_loss_fp32_mean_sd": [round(m32, 4), round(s32, 4)], "final_loss_bf16_mean_sd": [round(m16, 4), round(s16, 4)],
               "paired_rel_diff_bf16_minus_fp32_pct": {"mean": round(100 * mr, 2), "sd": round(100 * sr, 2), "standard_error": round(100 * se, 2),
                                                       "t": round(mr / se, 2) if se > 0 else None},
               "repeat_noise_pct": {"fp32_mean_abs": round(100 * sum(rep32) / len(rep32), 2), "bf16_mean_abs": round(100 * sum(rep16) / len(rep16), 2)},
               "first_loss_rel_diff_pct_max": round(100 * max(abs(r["bf16"]["first"] - r["fp32"]["first"]) / r["fp32"]["first"] for r in rows), 3)}
    print(json.dumps(summary))
    if a.out:
        json.dump({"what": __doc__.split("\n\n")[0], "summary": summary, "runs": rows}, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()

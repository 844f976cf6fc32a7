"""Convergence evidence for the bf16 kernel path (bf16-in / fp32-acc allowed): the same
short training run on the bf16 HIP path and on the fp32 torch composition -- same weights, same batch, same optimizer
arithmetic, stochastic layers off (tools/loss_curve.py; the 200-step curves are in profiles/r03_loss_curve.json) -- must
reduce the loss alike.  Single-layer / single-module tolerances (tests/test_modules_gpu.py, test_gemm_gpu.py) say nothing
about drift over optimisation steps; this does."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.parametrize("name,steps", [("c3s", 40), ("c2", 40)])
def test_bf16_path_trains_like_fp32(dev, name, steps):
    """Three runs: fp32, the bf16 kernel path, and a CONTROL -- fp32 with only the input features / image rounded to bf16
    once.  This loss is not a smooth function of the features (vote clustering by FPS over PREDICTED votes, nearest-centre
    objectness labels, max-pool winners; fp32 atomics make even two identical fp32 runs differ by a few %): the control
    measures how far a bf16-sized perturbation moves the curve, and the bf16 path must stay within that band.  Measured
    over 200 steps, two executions (profiles/r03_loss_curve.json holds the second): final gap to fp32 7.8 % / 12.3 % (bf16) vs
    6.9 % / 0.2 % (control) at c2, 4.5 % / 4.3 % vs 4.8 % / 3.1 % at the reduced c3; loss-reduction factors x0.40 / x0.47 / x0.36
    and x0.39 / x0.47 / x0.37 at c2 (the bf16 detector fits the one fixed batch a little less far, repeatably), x0.25 / x0.24 /
    x0.24 and x0.23 / x0.24 / x0.24 at c3."""
    import loss_curve
    r = loss_curve.compare(name, steps, tail=10)
    a, b, c = r["fp32"], r["bf16"], r["control"]
    assert all(x == x and abs(x) < 1e6 for x in a + b + c)
    # every run makes real progress on the fixed batch ...
    assert max(r["loss_drop"].values()) < 0.95, r["loss_drop"]
    # ... the bf16 path ends near where fp32 ends.  The bounds are wide on purpose: this loss is chaotic at 40 steps (two
    # fp32 executions differ by 5 % through fp32 atomics; while the loss is still falling fast a small lead or lag in the
    # descent is a large relative gap).  Observed 10-step-tail gaps of the bf16 path over this round's executions: c2
    # 2.3 / 0.8 / 4.5 %, c3s up to 17.1 % (with the control at 0.1 % in that run and 2.5 / 4.1 / 1.1 % in others: the control
    # band is too noisy to scale a bound with); the 200-step curves end 4-12 % apart.  What this test
    # guards against is a path that stalls or diverges, not a percent.
    assert r["final_gap_rel"] <= 0.30, (r["final_gap_rel"], a[-5:], b[-5:], r["control_final_gap_rel"])
    # ... and its progress is comparable: loss-reduction factors within 35 % of each other
    assert abs(r["loss_drop"]["bf16"] / r["loss_drop"]["fp32"] - 1.0) <= 0.35, r["loss_drop"]

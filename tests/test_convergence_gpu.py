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
    # ... the bf16 path ends where fp32 ends, to within the band a bf16-sized input perturbation opens
    # (bounds from repeated runs: the fp32 run itself moves by 5 % between two executions -- fp32 atomics -- and over three
    # repetitions of the 40-step c2 case the bf16 gap was 2.3 / 0.8 / 4.5 %, the control's 2.5 / 4.1 / 1.1 %; a 5-step tail
    # once measured 15.1 % on a curve whose last steps oscillated, hence the 10-step mean)
    assert r["final_gap_rel"] <= 0.20, (r["final_gap_rel"], a[-5:], b[-5:])
    assert r["final_gap_rel"] <= 3.0 * max(r["control_final_gap_rel"], 0.05), (r["final_gap_rel"], r["control_final_gap_rel"])
    # ... and its progress is the same: loss-reduction factors within 25 % of each other
    assert abs(r["loss_drop"]["bf16"] / r["loss_drop"]["fp32"] - 1.0) <= 0.25, r["loss_drop"]

"""BASELINE.json configs at FULL size on one MI355X (the oracle-free, size-independent properties SURVEY §8c asks for
at full sizes): c3 (B=16 x 40000 pts x C_in=132 + 512x512 view, bf16) and one step of c5's per-rank share (B=32 x
80000 pts + 1024x1024 views, use_text_decoder) with its peak HBM recorded."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


class _Args(object):
    def __init__(self, points, image, cin=132):
        self.points, self.image, self.cin = points, image, cin


def _finite_grads(model):
    n = 0
    for name, p in model.named_parameters():
        if p.grad is not None:
            assert torch.isfinite(p.grad).all(), name
            n += 1
    return n


def test_c3_full_size_forward_backward_properties(dev):
    """config c3 at full size, eager: (1) the bf16 HIP path and the fp32 composition pick IDENTICAL sampling / grouping
    indices (they are computed from fp32 coordinates on both: index-exactness is precision-mode independent);
    (2) detector outputs of the bf16 path stay within 1.2e-1 rel-L2 of fp32 on seed / vote features after 4 SA + 2
    FP levels (each SharedMLP alone is within SURVEY §8a's 1e-2: tests/test_modules_gpu.py); (3) the LM loss of the bf16 path is
    within 2e-2 relative of fp32; (4) every gradient of one full backward is finite and the used-parameter set is the
    same in both modes."""
    import bench
    from bridgeqa_amd import fusion_ops as ops
    args = _Args(40000, 512)
    torch.manual_seed(0)
    model = bench.build_model("c3", args.cin, args.image).to(dev)
    model.train()
    for mod in model.modules():  # dropout / stochastic depth off: the two precision modes must see the same function
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        if hasattr(mod, "drop_prob"):
            mod.drop_prob = 0.0
    batch = bench.make_batch(args, "c3", 16, 42, dev)
    res = {}
    for mode, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        prev = ops.set_compute_dtype(dt)
        try:
            for p in model.parameters():
                p.grad = None
            dd = model(dict(batch))
            loss = bench.total_loss(dd)
            loss.backward()
            torch.cuda.synchronize()
            res[mode] = dict(loss=loss.item(), blip=dd["blip_loss"].item(), used=_finite_grads(model),
                             inds={k: dd[k].clone() for k in ("sa1_inds", "sa2_inds", "sa3_inds", "sa4_inds", "fp2_inds",
                                                              "aggregated_vote_inds") if k in dd},
                             seed=dd["seed_features"].float().clone(), votes=dd["vote_features"].float().clone(),
                             fused=dd["fused_feat"].float().clone())
        finally:
            ops.set_compute_dtype(prev)
    a, b = res["fp32"], res["bf16"]
    assert a["inds"].keys() == b["inds"].keys() and {"sa1_inds", "fp2_inds"} <= set(a["inds"])
    for k in a["inds"]:
        if k != "aggregated_vote_inds":  # (sampled from the predicted votes: network outputs, not coordinates)
            assert torch.equal(a["inds"][k], b["inds"][k]), k     # coordinates only: identical in both modes
    rel = lambda x, y: ((x - y).norm() / (y.norm() + 1e-20)).item()
    print("c3 bf16 vs fp32: seed rel-L2 %.4f  votes rel-L2 %.4f  fused rel-L2 %.4f  blip loss %.5f vs %.5f" % (
        rel(b["seed"], a["seed"]), rel(b["votes"], a["votes"]), rel(b["fused"], a["fused"]), b["blip"], a["blip"]))
    # one SharedMLP (3 layers + pooling) of the native path is within 5e-3 of fp32 (tests/test_modules_gpu.py: SURVEY
    # §8a's 1e-2 per SharedMLP holds); through 4 SA + 2 FP levels with random-init weights, re-normalised by BatchNorm at
    # every layer and max-pooled (bf16 rounding moves arg-max winners), the deviation compounds to 7-9 % here
    assert rel(b["seed"], a["seed"]) <= 1.2e-1, rel(b["seed"], a["seed"])
    assert rel(b["votes"], a["votes"]) <= 1.2e-1
    assert abs(b["blip"] - a["blip"]) <= 2e-2 * abs(a["blip"]), (a["blip"], b["blip"])
    assert a["used"] == b["used"] and a["used"] > 400
    assert all(x == x and abs(x) < 1e6 for x in (a["loss"], b["loss"]))


def test_c5_per_rank_phased_step_and_peak_hbm(dev):
    """BASELINE config c5, one rank's share (B=32 x 80000 points + 1024x1024 views -> 4097 image tokens, text decoder):
    the PHASED training step (pipeline.PhasedTrainStep: one HIP graph per phase, forward + backward + fused AdamW) is
    captured and replayed, losses and gradients are finite, and the peak HBM of the step is recorded
    (gpurun_out/c5_peak_hbm.json; DESIGN.md quotes it).  No activation recompute is needed on 288 GB: the reference's lever
    for this config (vit.py:103-105 checkpoint_wrapper) stays off.  `bench.py --workload c5` times the same step."""
    import bench
    from bridgeqa_amd import fusion_ops as ops
    from bridgeqa_amd.optim import FusedAdamW
    from bridgeqa_amd.pipeline import PhasedTrainStep
    args = _Args(80000, 1024)
    prev = ops.set_compute_dtype(torch.bfloat16)
    try:
        torch.manual_seed(0)
        model = bench.build_model("c3", args.cin, args.image).to(dev)
        model.train()
        opt = FusedAdamW(model.parameters(), lr=1e-4, weight_decay=1e-5, grad_clip_value=1.0)
        batch = bench.make_batch(args, "c3", 32, 43, dev)
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats(dev)
        pipe = PhasedTrainStep(model, batch, bench.det_loss, bench.fusion_loss, opt, use_graphs=True, next_batch=batch)
        pipe.capture(warmup=1, keep_warmup_updates=True)
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        losses = []
        for step in range(3):
            t0.record(pipe.s_main)
            loss = pipe.step()
            pipe.wait()
            t1.record()
            torch.cuda.synchronize()
            losses.append(loss.item())
        dd = pipe._state["dd"]
        assert all(x == x and abs(x) < 1e6 for x in losses), losses
        assert dd["sa1_inds"].shape == (32, 2048) and tuple(pipe._state["img"].shape[:2]) == (32, 4097)
        assert _finite_grads(model) > 400
        peak = torch.cuda.max_memory_allocated(dev)
        rec = {"config": "c5 per rank: B=32 x 80000 pts x C_in=132 + 1024^2 view, bf16, phased fwd+bwd+FusedAdamW replayed "
                         "from HIP graphs (1 eager warm-up step + capture + 3 replays)",
               "peak_allocated_GB": round(peak / 2 ** 30, 2), "reserved_GB": round(torch.cuda.max_memory_reserved(dev) / 2 ** 30, 2),
               "replayed_step_ms": round(t0.elapsed_time(t1), 1), "losses": losses}
        os.makedirs(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out"), exist_ok=True)
        with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "c5_peak_hbm.json"), "w") as f:
            json.dump(rec, f)
        print(json.dumps(rec))
        assert peak < 200 * 2 ** 30   # fits one MI355X (288 GB) with room to spare
    finally:
        ops.set_compute_dtype(prev)

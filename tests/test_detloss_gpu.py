"""csrc/detloss.hip (bq_det_loss_fwd / bq_det_loss_bwd behind loss_helper.get_detection_loss): the detection loss and its
gradient in two launches against (a) the reference's own lib/loss_helper.py through tests/golden/det_loss.npz and (b) the
torch composition of bridgeqa_amd/loss_helper.py -- every term, the labels / masks / assignments exactly, the gradient
w.r.t. every network output -- in the three layouts the kernel reads: separate contiguous tensors, separate channel slices
(graphed.wrap_loss' leaves), slices of one head output inside the autograd graph (the training step)."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TERMS = ("vote_loss", "objectness_loss", "center_loss", "heading_cls_loss", "heading_reg_loss", "size_cls_loss", "size_reg_loss",
         "sem_cls_loss")
W = dict(vote_loss=1.0, objectness_loss=0.5, box_loss=1.0, sem_cls_loss=0.1)


def test_fused_detection_loss_vs_reference_golden(golden, dev):
    from bridgeqa_amd import loss_helper as lh
    from test_loss_cpu import load
    d, want, cfg = load(golden, dev)
    diff = ("vote_xyz", "center", "objectness_scores", "heading_scores", "heading_residuals_normalized", "size_scores",
            "size_residuals_normalized", "sem_cls_scores")
    for k in diff:
        d[k].requires_grad_(True)
    assert lh._fused_ok(d) is False          # fused, separate tensors
    loss, dd = lh.get_detection_loss(d, cfg, loss_weights=W)
    close = lambda a, b: torch.allclose(a.detach().cpu().float(), b.float(), rtol=1e-5, atol=1e-6)
    for k in TERMS:
        assert close(dd[k], want[k]), (k, dd[k].item(), want[k].item())
    for k in ("objectness_label", "objectness_mask", "object_assignment"):
        assert torch.equal(dd[k].cpu(), want[k]) and dd[k].dtype == want[k].dtype, k
    assert close(loss, want["total_x10"])
    loss.backward()
    got = {k: d[k].grad.clone() for k in diff}
    # the torch composition on the same inputs
    prev, lh.FUSED_DET_LOSS[0] = lh.FUSED_DET_LOSS[0], False
    try:
        for k in diff:
            d[k].grad = None
        loss2, dd2 = lh.get_detection_loss(d, cfg, loss_weights=W)
        loss2.backward()
    finally:
        lh.FUSED_DET_LOSS[0] = prev
    for k in diff:
        assert torch.allclose(got[k], d[k].grad, rtol=1e-5, atol=1e-7), (k, (got[k] - d[k].grad).abs().max().item())
    for k in ("pos_ratio", "neg_ratio"):
        assert torch.allclose(dd[k], dd2[k], atol=1e-7)


def _case(dev, B=4, S=1024, K=256, G=128, N=5000, NH=1, NS=18, NC=18, seed=0):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.rand(*s, generator=g)
    room = torch.tensor([8.0, 8.0, 3.0])
    gt = torch.full((B, G, 3), -100.0)
    gt[:, :8] = r(B, 8, 3) * room
    bm = torch.zeros(B, G); bm[:, :8] = 1
    seed_xyz = r(B, S, 3) * room
    agg = gt[:, :8].repeat(1, K // 8, 1) + 0.5 * torch.randn(B, K, 3, generator=g)   # near, grey-zone and far proposals
    C = 2 + 3 + 2 * NH + 4 * NS + NC
    d = dict(seed_xyz=seed_xyz, vote_xyz=seed_xyz + 0.3 * torch.randn(B, S, 3, generator=g),
             seed_inds=torch.randint(0, N, (B, S), generator=g).int(), vote_label=torch.randn(B, N, 9, generator=g),
             vote_label_mask=(r(B, N) > 0.5).long(), center_label=gt, box_label_mask=bm, aggregated_vote_xyz=agg,
             heading_class_label=torch.randint(0, NH, (B, G), generator=g), heading_residual_label=0.3 * torch.randn(B, G, generator=g),
             size_class_label=torch.randint(0, NS, (B, G), generator=g), size_residual_label=0.5 * torch.randn(B, G, 3, generator=g),
             sem_cls_label=torch.randint(0, NC, (B, G), generator=g))
    net = 2.0 * torch.randn(B, K, C, generator=g)      # (Huber's linear branch and both CE regimes are hit)
    cfg = types.SimpleNamespace(num_heading_bin=NH, num_size_cluster=NS, num_class=NC,
                                mean_size_arr=0.5 + np.random.RandomState(seed).rand(NS, 3))
    return {k: v.to(dev) for k, v in d.items()}, net.to(dev), cfg


def _decode(net, d, NH, NS):
    """proposal_module.decode_scores' slicing"""
    B, K = net.shape[:2]
    o, out = 0, dict(d)
    out["objectness_scores"] = net[:, :, o:o + 2]; o += 2
    out["center"] = d["aggregated_vote_xyz"] + net[:, :, o:o + 3]; o += 3
    out["heading_scores"] = net[:, :, o:o + NH]; o += NH
    out["heading_residuals_normalized"] = net[:, :, o:o + NH]; o += NH
    out["size_scores"] = net[:, :, o:o + NS]; o += NS
    out["size_residuals_normalized"] = net[:, :, o:o + NS * 3].view(B, K, NS, 3); o += NS * 3
    out["sem_cls_scores"] = net[:, :, o:]
    return out


@pytest.mark.parametrize("NH,VF", [(1, 1), (12, 2)])
def test_fused_detection_loss_equals_the_torch_composition_in_every_layout(dev, NH, VF):
    from bridgeqa_amd import loss_helper as lh
    d, net, cfg = _case(dev, NH=NH)
    if VF > 1:
        d["vote_xyz"] = d["vote_xyz"].repeat_interleave(VF, 1) + 0.1 * torch.randn(d["vote_xyz"].shape[0], d["vote_xyz"].shape[1] * VF, 3, device=dev)
    NS = cfg.num_size_cluster
    tw = torch.tensor([1.0, 0.5, 1.3, 0.1, 0.7, 0.2, 1.1, 0.3], device=dev)   # distinct upstream gradients per term

    def run(fused, layout):
        vote = d["vote_xyz"].clone().requires_grad_(True)
        base = net.clone().requires_grad_(True)
        dd = _decode(base, dict(d, vote_xyz=vote), NH, NS)
        leaves = {}
        if layout == "slices":      # independent leaves with the slices' strides (graphed.wrap_loss)
            for k in ("objectness_scores", "heading_scores", "heading_residuals_normalized", "size_scores",
                      "size_residuals_normalized", "sem_cls_scores", "center"):
                leaves[k] = dd[k] = dd[k].detach().requires_grad_(True)
        elif layout == "contiguous":
            for k in ("objectness_scores", "heading_scores", "heading_residuals_normalized", "size_scores",
                      "size_residuals_normalized", "sem_cls_scores", "center"):
                leaves[k] = dd[k] = dd[k].detach().contiguous().requires_grad_(True)
        prev, lh.FUSED_DET_LOSS[0] = lh.FUSED_DET_LOSS[0], fused
        try:
            if fused:
                mode = lh._fused_ok(dd)
                assert (mode is not None) and (bool(mode) == (layout == "packed")), (layout, mode)
            _, dd = lh.get_detection_loss(dd, cfg, loss_weights=W)
        finally:
            lh.FUSED_DET_LOSS[0] = prev
        terms = torch.stack([dd[k] for k in TERMS])
        (terms * tw).sum().backward()
        grads = {"vote_xyz": vote.grad}
        grads.update({k: v.grad for k, v in leaves.items()} if leaves else {"net": base.grad})
        return terms.detach(), {k: dd[k] for k in ("objectness_label", "objectness_mask", "object_assignment", "pos_ratio", "neg_ratio")}, grads

    for layout in ("packed", "slices", "contiguous"):
        want_t, want_l, want_g = run(False, layout)
        got_t, got_l, got_g = run(True, layout)
        assert torch.allclose(got_t, want_t, rtol=2e-5, atol=1e-6), (layout, got_t, want_t)
        for k in ("objectness_label", "objectness_mask", "object_assignment"):
            assert torch.equal(got_l[k], want_l[k]), (layout, k)
        assert 0.02 < want_l["pos_ratio"].item() < 0.9 and want_l["neg_ratio"].item() > 0.02   # (all three label classes occur)
        for k in ("pos_ratio", "neg_ratio"):
            assert torch.allclose(got_l[k], want_l[k], atol=1e-6)
        assert set(got_g) == set(want_g)
        for k in want_g:
            err = (got_g[k] - want_g[k]).abs().max().item() / (want_g[k].abs().max().item() + 1e-12)
            assert err < 2e-5, (layout, k, err)


def test_fused_detection_loss_is_two_launches_and_reproducible(dev):
    """bitwise identical terms and gradients across executions (fixed reduction order, no atomics)"""
    from bridgeqa_amd import loss_helper as lh
    d, net, cfg = _case(dev, seed=3)
    outs = []
    for _ in range(2):
        base = net.clone().requires_grad_(True)
        vote = d["vote_xyz"].clone().requires_grad_(True)
        dd = _decode(base, dict(d, vote_xyz=vote), 1, cfg.num_size_cluster)
        loss, dd = lh.get_detection_loss(dd, cfg, loss_weights=W)
        loss.backward()
        outs.append((loss.detach().clone(), base.grad.clone(), vote.grad.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)

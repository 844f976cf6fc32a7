"""bridgeqa_amd/loss_helper.py + nn_distance.py against golden vectors produced by the reference's own
lib/loss_helper.py (oracle/gen_golden_loss.py): every loss term, the labels / masks / assignments, the weighted total."""
import types

import numpy as np
import pytest
import torch


def load(golden, device):
    g = golden("det_loss.npz")
    d = {k[3:]: torch.from_numpy(g[k]).to(device) for k in g if k.startswith("in_")}
    out = {k[4:]: torch.from_numpy(g[k]) for k in g if k.startswith("out_")}
    nh, ns, nc = (int(x) for x in g["dims"])
    cfg = types.SimpleNamespace(num_heading_bin=nh, num_size_cluster=ns, num_class=nc, mean_size_arr=g["mean_size_arr"])
    return d, out, cfg


def run_loss_golden(golden, device, tol):
    from bridgeqa_amd import loss_helper as lh
    d, want, cfg = load(golden, device)
    for k in ("vote_xyz", "aggregated_vote_xyz", "center", "objectness_scores", "size_residuals_normalized"):
        d[k].requires_grad_(True)
    loss, dd = lh.get_detection_loss(d, cfg, loss_weights=dict(vote_loss=1.0, objectness_loss=0.5, box_loss=1.0,
                                                                sem_cls_loss=0.1))
    close = lambda a, b: torch.allclose(a.detach().cpu().float(), b.float(), rtol=tol, atol=tol)
    for k in ("vote_loss", "objectness_loss", "center_loss", "heading_cls_loss", "heading_reg_loss", "size_cls_loss",
              "size_reg_loss", "sem_cls_loss"):
        assert close(dd[k], want[k]), (k, dd[k].item(), want[k].item())
    assert torch.equal(dd["objectness_label"].cpu(), want["objectness_label"])
    assert torch.equal(dd["objectness_mask"].cpu(), want["objectness_mask"])
    assert torch.equal(dd["object_assignment"].cpu(), want["object_assignment"])
    assert close(loss, want["total_x10"])
    loss.backward()
    for k in ("vote_xyz", "center", "objectness_scores", "size_residuals_normalized"):
        assert d[k].grad is not None and torch.isfinite(d[k].grad).all() and d[k].grad.abs().sum() > 0


def test_detection_losses_vs_reference_golden(golden):
    run_loss_golden(golden, torch.device("cpu"), 1e-6)


def test_nn_distance_and_huber_definitions():
    from bridgeqa_amd.nn_distance import huber_loss, nn_distance
    g = torch.Generator().manual_seed(0)
    a, b = torch.rand(2, 5, 3, generator=g), torch.rand(2, 6, 3, generator=g)
    for kw in ({}, {"l1": True}, {"l1smooth": True, "delta": 0.3}):
        d1, i1, d2, i2 = nn_distance(a, b, **kw)
        diff = a[:, :, None] - b[:, None]
        if kw.get("l1"):
            full = diff.abs().sum(-1)
        elif kw.get("l1smooth"):
            e = diff.abs(); q = e.clamp(max=0.3)
            full = (0.5 * q ** 2 + 0.3 * (e - q)).sum(-1)
        else:
            full = (diff ** 2).sum(-1)
        assert torch.allclose(d1, full.min(2)[0]) and torch.equal(i1, full.min(2)[1])
        assert torch.allclose(d2, full.min(1)[0]) and torch.equal(i2, full.min(1)[1])
    x = torch.tensor([-2.0, -0.5, 0.0, 0.5, 2.0])
    assert torch.allclose(huber_loss(x, 1.0), torch.tensor([1.5, 0.125, 0.0, 0.125, 1.5]))

"""Proposal post-processing on the device (csrc/nms.hip + bridgeqa_amd/ap_helper.py) against golden
vectors produced by the reference's own lib/ap_helper.parse_predictions (oracle/gen_golden_nms.py): `pred_mask` exact,
`batch_pred_map_cls` entry by entry, in five config_dict variants (3-D NMS per class / class-agnostic / old-type overlap,
bird's-eye 2-D NMS, empty-box removal on and off, per-class proposals on and off)."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

VARIANTS = (
    dict(remove_empty_box=True, use_3d_nms=True, nms_iou=0.25, use_old_type_nms=False, cls_nms=True, per_class_proposal=True, conf_thresh=0.05),
    dict(remove_empty_box=False, use_3d_nms=True, nms_iou=0.25, use_old_type_nms=False, cls_nms=False, per_class_proposal=False, conf_thresh=0.3),
    dict(remove_empty_box=False, use_3d_nms=False, nms_iou=0.3, use_old_type_nms=False, cls_nms=False, per_class_proposal=False, conf_thresh=0.05),
    dict(remove_empty_box=True, use_3d_nms=True, nms_iou=0.2, use_old_type_nms=True, cls_nms=False, per_class_proposal=False, conf_thresh=0.05),
    dict(remove_empty_box=False, use_3d_nms=False, nms_iou=0.25, use_old_type_nms=True, cls_nms=False, per_class_proposal=True, conf_thresh=0.5),
)


@pytest.mark.parametrize("v", range(5))
def test_parse_predictions_vs_reference_golden(golden, dev, v):
    from bridgeqa_amd.ap_helper import parse_predictions
    g = golden("nms.npz")
    pre = "v%d_in_" % v
    end = {k[len(pre):]: torch.from_numpy(g[k]).to(dev) for k in g if k.startswith(pre)}
    cfg = types.SimpleNamespace(num_heading_bin=int(g["v%d_NH" % v]), num_class=18, mean_size_arr=g["v%d_mean_size_arr" % v])
    out = parse_predictions(end, dict(VARIANTS[v], dataset_config=cfg))
    assert np.array_equal(end["pred_mask"], g["v%d_pred_mask" % v])
    assert torch.equal(end["pred_mask_device"].cpu(), torch.from_numpy(g["v%d_pred_mask" % v]).bool())
    assert len(out) == end["center"].shape[0]
    for i, lst in enumerate(out):
        cls, score, corners = g["v%d_b%d_cls" % (v, i)], g["v%d_b%d_score" % (v, i)], g["v%d_b%d_corners" % (v, i)]
        assert len(lst) == len(cls)
        assert [c for c, _, _ in lst] == cls.tolist()
        assert np.allclose([s for _, _, s in lst], score, rtol=1e-5, atol=1e-7)
        assert np.allclose(np.array([c for _, c, _ in lst]).reshape(-1, 8, 3), corners, rtol=0, atol=2e-5)


def test_box_point_count_vs_brute_force(dev):
    from bridgeqa_amd import _ext
    g = torch.Generator().manual_seed(5)
    B, N, K = 2, 5000, 40
    pc = torch.cat([torch.rand(B, N, 3, generator=g) * 4, torch.randn(B, N, 4, generator=g)], -1).to(dev)
    center = (torch.rand(B, K, 3, generator=g) * 4).to(dev)
    size = (torch.rand(B, K, 3, generator=g) * 1.5 + 0.1).to(dev)
    heading = ((torch.rand(B, K, generator=g) - 0.5) * 6).to(dev)
    got = _ext.box_point_count(pc, center, size, heading)
    d = pc[:, None, :, :3] - center[:, :, None, :]                      # (B, K, N, 3)
    c, s = torch.cos(heading)[..., None], torch.sin(heading)[..., None]
    x, y, z = c * d[..., 0] - s * d[..., 2], d[..., 1], s * d[..., 0] + c * d[..., 2]
    h = size[:, :, None, :] / 2
    want = ((x.abs() <= h[..., 0]) & (y.abs() <= h[..., 1]) & (z.abs() <= h[..., 2])).sum(-1)
    assert (got.long() - want).abs().max().item() <= 1      # (a point within an ulp of a face may flip)
    assert torch.equal(_ext.box_point_count(pc, center, size, heading, cap=5), got.clamp(max=5))
    with pytest.raises(RuntimeError):
        _ext.box_point_count(pc.cpu(), center, size, heading)


def test_nms_edge_cases(dev):
    from bridgeqa_amd import _ext
    # identical boxes: only the best survives; disjoint boxes: all survive; invalid boxes never picked nor suppressing
    box = torch.tensor([[[0, 0, 0, 1, 1, 1], [0, 0, 0, 1, 1, 1], [5, 5, 5, 6, 6, 6], [0, 0, 0, 1, 1, 1.0]]], device=dev)
    score = torch.tensor([[0.3, 0.9, 0.1, 0.95]], device=dev)
    keep = _ext.nms(box, score, 0.25)
    assert keep.tolist() == [[False, False, True, True]]
    keep = _ext.nms(box, score, 0.25, valid=torch.tensor([[1, 1, 1, 0]], device=dev, dtype=torch.uint8))
    assert keep.tolist() == [[False, True, True, False]]
    cls = torch.tensor([[0, 1, 0, 1]], device=dev)
    keep = _ext.nms(box, score, 0.25, cls=cls, same_cls=True)
    assert keep.tolist() == [[True, False, True, True]]      # class 0 box survives next to the class-1 pair
    with pytest.raises(RuntimeError):
        _ext.nms(torch.zeros(1, 1025, 6, device=dev), torch.zeros(1, 1025, device=dev), 0.25)

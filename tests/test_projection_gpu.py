"""bridgeqa_amd.projection against the reference's own lib/projection.py ProjectionHelper + the aggregation of
scripts/project_multiview_features.py (tests/golden/projection.npz, made by oracle/gen_golden_projection.py)."""
import os
import pickle
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "projection.npz")
pytestmark = pytest.mark.gpu


def setup(dev):
    from bridgeqa_amd.projection import ProjectionHelper
    z = np.load(GOLD)
    dmin, dmax, acc = (float(x) for x in z["limits"])
    helper = ProjectionHelper(z["intrinsics"].tolist(), dmin, dmax, [int(x) for x in z["image_dims"]], acc, device=dev)
    return z, helper


def features(z, F):
    from gen_golden_projection import make_features  # the seeded generator of the golden (integers, exact in fp32)
    return make_features(int(z["feat_seed"]), F)


def test_single_frame_interface_matches_reference(dev):
    z, helper = setup(dev)
    pts = torch.from_numpy(z["points"]).to(dev)
    F = z["poses"].shape[0]
    for f in range(F):
        got = helper.compute_projection(pts, torch.from_numpy(z["depths"][f]).to(dev), torch.from_numpy(z["poses"][f]).to(dev))
        if "f%d_none" % f in z.files:
            assert got is None
            continue
        assert got is not None
        assert torch.equal(got[0].cpu(), torch.from_numpy(z["f%d_indices_3d" % f])), f
        assert torch.equal(got[1].cpu(), torch.from_numpy(z["f%d_indices_2d" % f])), f
    # ProjectionHelper.project on frame 0's mapping
    feats = features(z, F).to(dev)
    got = helper.compute_projection(pts, torch.from_numpy(z["depths"][0]).to(dev), torch.from_numpy(z["poses"][0]).to(dev))
    out = helper.project(feats[0], got[0], got[1], pts.shape[0])
    assert torch.equal(out.cpu(), torch.from_numpy(z["project_f0"].astype(np.float32)))


def test_batched_frame_records_equal_the_per_frame_calls(dev):
    z, helper = setup(dev)
    g = torch.Generator().manual_seed(1)
    poses = torch.from_numpy(z["poses"])[torch.randint(0, 6, (40,), generator=g)].clone()
    poses[:, :3, 3] += torch.randn(40, 3, generator=g)
    one_by_one = torch.stack([helper._frame_record(poses[f]) for f in range(40)])
    assert torch.equal(helper._frame_records(poses), one_by_one)


@pytest.mark.parametrize("maxpool", [False, True])
def test_scene_aggregation_matches_reference(dev, maxpool, tmp_path):
    from bridgeqa_amd.projection import save_point_features
    z, helper = setup(dev)
    F = z["poses"].shape[0]
    out = helper.project_scene(z["points"], z["depths"], z["poses"], features(z, F), maxpool=maxpool)
    want = z["point_features_%s" % ("maxpool" if maxpool else "first")].astype(np.float32)
    assert out.shape == want.shape and out.dtype == torch.float32
    assert np.array_equal(out.cpu().numpy(), want)
    path = os.path.join(str(tmp_path), "scene0000_00.pkl")
    save_point_features(path, out)
    with open(path, "rb") as f:
        back = pickle.load(f)        # lib/dataset.py:410-411
    assert isinstance(back, np.ndarray) and np.array_equal(back, want)


def test_pixel_map_properties_at_scene_size(dev):
    """a ScanNet-sized scene (150 k vertices x 120 frames): every mapped pixel lies inside the image, its depth passes the
    reference's three depth tests, and a second run gives the same map"""
    z, helper = setup(dev)
    g = torch.Generator().manual_seed(0)
    N, F = 150000, 120
    pts = (torch.rand(N, 3, generator=g) * torch.tensor([6.0, 5.0, 3.0])).to(dev)
    poses = torch.from_numpy(z["poses"])[torch.randint(0, 4, (F,), generator=g)].clone()
    poses[:, :3, 3] += torch.randn(F, 3, generator=g) * 0.2
    depths = torch.rand(F, 32, 41, generator=g) * 4.5
    pix = helper.project_frames(pts, depths, poses)
    assert torch.equal(pix, helper.project_frames(pts, depths, poses))
    seen = pix >= 0
    assert int(seen.sum()) > 0 and int(pix.max()) < 32 * 41 and int(pix.min()) >= -1
    d = depths.to(dev).reshape(F, -1).gather(1, pix.clamp(min=0).long())[seen]
    assert bool(((d >= helper.depth_min) & (d <= helper.depth_max)).all())


def test_edge_cases_no_frame_sees_anything_and_single_point(dev):
    """a scene no camera looks at -> an all -1 map, all-zero features, `compute_projection` returns None (the reference's
    early `return None`s, projection.py:222,236,243); one point, one frame, 64 channels"""
    z, helper = setup(dev)
    far = torch.from_numpy(z["points"]) + 100.0
    F = z["poses"].shape[0]
    pix = helper.project_frames(far, z["depths"], z["poses"])
    assert pix.shape == (F, far.shape[0]) and int(pix.max()) == -1
    out = helper.project_scene(far, z["depths"], z["poses"], features(z, F), maxpool=True)
    assert float(out.abs().max()) == 0.0
    assert helper.compute_projection(far.to(dev), torch.from_numpy(z["depths"][0]).to(dev),
                                     torch.from_numpy(z["poses"][0]).to(dev)) is None
    # a single point that frame 0 does see, 64-channel features
    seen = torch.from_numpy(z["f0_indices_3d"])
    p = torch.from_numpy(z["points"])[int(seen[1])].reshape(1, 3)
    g = torch.Generator().manual_seed(2)
    feat = torch.randn(1, 64, 32, 41, generator=g)
    one = helper.project_scene(p, z["depths"][:1], z["poses"][:1], feat, maxpool=False)
    pixel = int(z["f0_indices_2d"][1])
    assert torch.equal(one.cpu()[0], feat[0].reshape(64, -1)[:, pixel])

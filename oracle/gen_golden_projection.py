"""Golden vectors for the offline multiview projection, produced by the REFERENCE's own
lib/projection.py ProjectionHelper (:5-276, pure torch, run on the CPU here) -- compute_projection per frame and project --
followed by the per-scene aggregation of scripts/project_multiview_features.py:155-202 (the script's `__main__` body is
not importable: its aggregation rule is restated in `aggregate_reference` below, statement for statement, on the
reference's own `PROJECTOR.project` outputs).

TEST INFRASTRUCTURE.  Usage:  python oracle/gen_golden_projection.py   -> tests/golden/projection.npz

Synthetic scene: points on the floor, the walls and a few boxes of a 6 x 5 x 3 m room; F camera poses inside it (ScanNet
convention: x right, y down, z forward; intrinsics of the 41 x 32 feature map, project_multiview_features.py:29-33); the
depth maps are the z-buffer of the points themselves at that resolution, so that the |depth - z| <= 0.05 test keeps the
front-most points of every pixel.  "ENet" features: small integers (exact in fp32, compressible), negative values
included, some pixels all-zero (the script's `== 0 ... 128` tests treat an all-zero feature vector as "not covered").
Features are regenerated in the test from the stored seed.
"""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, "tests", "golden", "projection.npz")
sys.path.insert(0, os.path.join(REPO, "oracle"))
from gen_golden_loss import install_shims  # noqa: E402

INTRINSICS = [[37.01983, 0, 20, 0], [0, 38.52470, 15.5, 0], [0, 0, 1, 0], [0, 0, 0, 1]]
IMAGE_DIMS = [41, 32]
DEPTH_MIN, DEPTH_MAX, ACCURACY = 0.1, 4.0, 0.05
C = 128


def make_scene(seed, n):
    g = np.random.RandomState(seed)
    pts = []
    k = n // 8
    pts.append(np.stack([g.rand(2 * k) * 6, g.rand(2 * k) * 5, np.zeros(2 * k)], 1))           # floor
    pts.append(np.stack([g.rand(k) * 6, np.zeros(k), g.rand(k) * 3], 1))                        # wall y = 0
    pts.append(np.stack([g.rand(k) * 6, np.full(k, 5.0), g.rand(k) * 3], 1))                    # wall y = 5
    pts.append(np.stack([np.zeros(k), g.rand(k) * 5, g.rand(k) * 3], 1))                        # wall x = 0
    pts.append(np.stack([np.full(k, 6.0), g.rand(k) * 5, g.rand(k) * 3], 1))                    # wall x = 6
    rest = n - 6 * k
    c = g.rand(4, 3) * [4, 3, 0.5] + [1, 1, 0.3]
    box = c[g.randint(0, 4, rest)] + (g.rand(rest, 3) - 0.5) * [0.8, 0.8, 0.6]
    pts.append(box)
    return np.concatenate(pts, 0).astype(np.float32)


def make_pose(pos, yaw, pitch):
    """camera_to_world of a camera at `pos` looking along the horizontal direction `yaw`, tilted down by `pitch`"""
    f = np.array([np.cos(yaw) * np.cos(pitch), np.sin(yaw) * np.cos(pitch), -np.sin(pitch)])    # forward (camera +z)
    r = np.array([np.sin(yaw), -np.cos(yaw), 0.0])                                              # right (camera +x)
    d = np.cross(f, r)                                                                          # down (camera +y)
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = r, d, f, pos
    return m.astype(np.float32)


def zbuffer(points, pose):
    w2c = np.linalg.inv(pose.astype(np.float64))
    cam = w2c @ np.concatenate([points.astype(np.float64), np.ones((len(points), 1))], 1).T
    z = cam[2]
    u = np.round(cam[0] * INTRINSICS[0][0] / z + INTRINSICS[0][2])
    v = np.round(cam[1] * INTRINSICS[1][1] / z + INTRINSICS[1][2])
    ok = (z > 0.05) & (u >= 0) & (u < IMAGE_DIMS[0]) & (v >= 0) & (v < IMAGE_DIMS[1])
    depth = np.zeros((IMAGE_DIMS[1], IMAGE_DIMS[0]), dtype=np.float32)
    best = np.full(IMAGE_DIMS[0] * IMAGE_DIMS[1], np.inf)
    pix = (v[ok] * IMAGE_DIMS[0] + u[ok]).astype(np.int64)
    np.minimum.at(best, pix, z[ok])
    depth.reshape(-1)[np.isfinite(best)] = best[np.isfinite(best)].astype(np.float32)
    return depth


def make_features(seed, F):
    g = torch.Generator().manual_seed(seed)
    feats = torch.randint(-4, 5, (F, C, IMAGE_DIMS[1], IMAGE_DIMS[0]), generator=g).float()
    dead = torch.rand(F, 1, IMAGE_DIMS[1], IMAGE_DIMS[0], generator=g) < 0.06        # all-zero feature vectors
    return feats * (~dead)


def aggregate_reference(projector, scene, frames, feats, maxpool):
    """scripts/project_multiview_features.py:171-198 on the reference's ProjectionHelper.project"""
    n = scene.shape[0]
    point_features = torch.zeros(n, C)
    for i, (f, p3, p2) in enumerate(frames):
        proj_feat = projector.project(feats[f], p3, p2, n).transpose(1, 0)
        if maxpool:
            feat_mask = ((proj_feat == 0).sum(1) != C).bool()
            point_mask = ((point_features == 0).sum(1) == C).bool()
            mask = point_mask * feat_mask
            point_features[mask] = proj_feat[mask]
            mask = ~point_mask * feat_mask
            point_features[mask] = torch.max(point_features[mask], proj_feat[mask])
        else:
            if i == 0:
                point_features = proj_feat
            else:
                mask = (point_features == 0).sum(1) == C
                point_features[mask] = proj_feat[mask]
    return point_features


def main():
    install_shims()
    from lib.projection import ProjectionHelper
    projector = ProjectionHelper(INTRINSICS, DEPTH_MIN, DEPTH_MAX, IMAGE_DIMS, ACCURACY, cuda=False, device=torch.device("cpu"))
    N, F, seed = 3000, 6, 11
    scene = make_scene(seed, N)
    poses = np.stack([make_pose([3.0, 2.5, 1.6], 0.3, 0.35), make_pose([1.0, 1.0, 1.5], 0.9, 0.25),
                      make_pose([5.0, 4.0, 1.4], 3.6, 0.3), make_pose([3.0, 4.5, 1.7], -1.4, 0.5),
                      make_pose([9.0, 9.0, 1.5], 0.5, 0.2),              # outside, looking away: no mapping at all
                      make_pose([2.0, 2.0, 2.2], 2.2, 0.9)])
    depths = np.stack([zbuffer(scene, p) for p in poses])
    depths[3, :6] = 0.0                                                   # a band of missing depth (below depth_min)
    feats = make_features(seed, F)
    save = dict(points=scene, poses=poses, depths=depths, feat_seed=np.array(seed), intrinsics=np.array(INTRINSICS, dtype=np.float64),
                image_dims=np.array(IMAGE_DIMS), limits=np.array([DEPTH_MIN, DEPTH_MAX, ACCURACY]))
    frames, counts = [], []
    for f in range(F):
        out = projector.compute_projection(torch.from_numpy(scene), torch.from_numpy(depths[f]), torch.from_numpy(poses[f]))
        if out is None:
            counts.append(0)
            save["f%d_none" % f] = np.array(1)
            continue
        p3, p2 = out
        counts.append(int(p3[0]))
        save["f%d_indices_3d" % f] = p3.numpy()
        save["f%d_indices_2d" % f] = p2.numpy()
        if int(p3[0]) > 0:
            frames.append((f, p3, p2))
    # one single-frame `project` (ProjectionHelper.project :235-255) kept on its own
    f0, p3, p2 = frames[0]
    save["project_f%d" % f0] = projector.project(feats[f0], p3, p2, N).numpy().astype(np.int8)
    for maxpool in (False, True):
        out = aggregate_reference(projector, scene, frames, feats, maxpool)
        assert float((out - out.round()).abs().max()) == 0 and float(out.abs().max()) < 100
        save["point_features_%s" % ("maxpool" if maxpool else "first")] = out.numpy().astype(np.int8)
        print("maxpool" if maxpool else "first", "points with features:", int(((out != 0).sum(1) > 0).sum()), "of", N)
    print("mappings per frame", counts)
    np.savez_compressed(OUT, **save)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()

"""Multiview projection of per-frame image features onto a scene's points -- mirror of the reference's lib/projection.py
`ProjectionHelper` (:5-276) and of the per-scene loop of scripts/project_multiview_features.py:103-202 that writes the
`enet_feats_maxpool/<scene>.pkl` files lib/dataset.py:408-412 reads (offline preprocessing).

Same class name, constructor and single-frame methods (`compute_projection` -> (indices_3d, indices_2d) | None with the
count in slot 0, `project`), so the reference script runs on it unchanged.  What changes underneath: the reference walks
the frames one by one and compacts boolean masks at every test (`.any()`, mask indexing, `nonzero`: a device->host
synchronisation each, ~8 per frame, hundreds of frames per scene), then aggregates the frames' features into the points
one frame at a time.  `project_scene` does a scene in two launches (csrc/projection.hip): every (frame, point) pair ->
pixel or -1, then every point walks its frames (first-fill or max-pool, the script's rule literally -- including its
"all-zero vector = not covered" tests).

Host-side, per frame, exactly as the reference computes them (fp32, the same torch calls, on the CPU: a few hundred 4x4
products): world-to-camera = torch.inverse(camera_to_world) (:212), the eight frustum corners (:58-82) and the six plane
normals (:84-130).
"""
import pickle

import numpy as np
import torch

from . import _ext


class ProjectionHelper(object):
    def __init__(self, intrinsic, depth_min, depth_max, image_dims, accuracy, cuda=True, device=None):
        self.intrinsic, self.depth_min, self.depth_max = intrinsic, depth_min, depth_max
        self.image_dims, self.accuracy, self.cuda = image_dims, accuracy, cuda
        if device is None:
            # (the reference takes cuda:<rank>; without an initialised process group that is the current device)
            device = torch.device("cuda", torch.cuda.current_device()) if cuda else torch.device("cpu")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("bridgeqa_amd.projection runs on the HIP kernels only (device %s)" % self.device)
        self._compute_corner_points()

    # ---- projection.py:24-56, on the host -----------------------------------------------------------------------------
    def depth_to_skeleton(self, ux, uy, depth):
        x = (ux - self.intrinsic[0][2]) / self.intrinsic[0][0]
        y = (uy - self.intrinsic[1][2]) / self.intrinsic[1][1]
        return torch.Tensor([depth * x, depth * y, depth])

    def skeleton_to_depth(self, p):
        x = (p[0] * self.intrinsic[0][0]) / p[2] + self.intrinsic[0][2]
        y = (p[1] * self.intrinsic[1][1]) / p[2] + self.intrinsic[1][2]
        return torch.Tensor([x, y, p[2]])

    def _compute_corner_points(self):
        w, h = self.image_dims[0] - 1, self.image_dims[1] - 1
        cp = torch.ones(8, 4)
        for i, (ux, uy, d) in enumerate(((0, 0, self.depth_min), (w, 0, self.depth_min), (w, h, self.depth_min),
                                         (0, h, self.depth_min), (0, 0, self.depth_max), (w, 0, self.depth_max),
                                         (w, h, self.depth_max), (0, h, self.depth_max))):
            cp[i][:3] = self.depth_to_skeleton(ux, uy, d)
        self.corner_points = cp  # (host copy: the frustum of a frame is 8 products of a 4x4 matrix)

    def compute_frustum_corners(self, camera_to_world):
        return torch.bmm(camera_to_world.repeat(8, 1, 1), self.corner_points.unsqueeze(2))

    def compute_frustum_normals(self, corner_coords):
        cc = corner_coords
        normals = cc.new(6, 3)
        for k, (a, b, c) in enumerate(((3, 1, 0), (2, 5, 1), (3, 6, 2), (0, 7, 3), (1, 4, 0), (6, 4, 5))):
            # plane k: cross(corner a - corner c, corner b - corner c) (projection.py:96-128)
            normals[k] = torch.cross(cc[a][:3].view(-1) - cc[c][:3].view(-1), cc[b][:3].view(-1) - cc[c][:3].view(-1), dim=0)
        return normals

    def _frame_record(self, camera_to_world):
        """the 40 floats of one frame for bq_project_points (include/bqhip_fusion.h), with the reference's own calls"""
        c2w = torch.as_tensor(camera_to_world, dtype=torch.float32).cpu()
        w2c = torch.inverse(c2w)
        cc = self.compute_frustum_corners(c2w)
        normals = self.compute_frustum_normals(cc)
        return torch.cat([w2c.reshape(16), normals.reshape(18), cc[2][:3].reshape(3), cc[4][:3].reshape(3)])

    def _frame_records(self, camera_to_worlds):
        """_frame_record of every frame in a few batched calls (the same LAPACK inverse / bmm / cross per matrix: batch
        elements are independent, the records are bit-identical -- tests/test_projection_gpu.py): 300 frames in ~1 ms
        instead of ~30 ms of per-frame Python"""
        c2w = torch.as_tensor(camera_to_worlds, dtype=torch.float32).cpu()
        F = c2w.shape[0]
        w2c = torch.inverse(c2w)
        cc = torch.bmm(c2w.repeat_interleave(8, 0), self.corner_points.repeat(F, 1).unsqueeze(2)).view(F, 8, 4)[:, :, :3]
        normals = torch.stack([torch.cross(cc[:, a] - cc[:, c], cc[:, b] - cc[:, c], dim=1)
                               for a, b, c in ((3, 1, 0), (2, 5, 1), (3, 6, 2), (0, 7, 3), (1, 4, 0), (6, 4, 5))], 1)
        return torch.cat([w2c.reshape(F, 16), normals.reshape(F, 18), cc[:, 2], cc[:, 4]], 1)

    # ---- a whole scene ---------------------------------------------------------------------------------------------------
    def project_frames(self, points, depths, camera_to_worlds):
        """points (N, 3), depths (F, H, W), camera_to_worlds (F, 4, 4) -> int32 (F, N) device tensor: the pixel index
        y * W + x that sees each point in each frame, -1 where it is not seen"""
        pts = torch.as_tensor(points, dtype=torch.float32).to(self.device).contiguous()
        dep = torch.as_tensor(depths, dtype=torch.float32).to(self.device).contiguous()
        poses = torch.as_tensor(camera_to_worlds, dtype=torch.float32).cpu()
        frames = self._frame_records(poses).to(self.device).contiguous()
        K = self.intrinsic
        return _ext.project_points(pts, dep, frames, self.image_dims, K[0][0], K[1][1], K[0][2], K[1][2], self.depth_min,
                                   self.depth_max, self.accuracy)

    def project_scene(self, points, depths, camera_to_worlds, features, maxpool=True):
        """scripts/project_multiview_features.py:150-198 for one scene: features (F, C, H, W) (the frames' ENet maps, in
        frame order) -> (N, C) float32 point features on the device"""
        pix = self.project_frames(points, depths, camera_to_worlds)
        feat = torch.as_tensor(features, dtype=torch.float32).to(self.device)
        F, C = feat.shape[0], feat.shape[1]
        feat = feat.reshape(F, C, -1).permute(0, 2, 1).contiguous()   # pixel-major: one contiguous row per (frame, pixel)
        return _ext.fuse_point_features(pix, feat, maxpool)

    # ---- the reference's single-frame interface -------------------------------------------------------------------------
    def compute_projection(self, points, depth, camera_to_world):
        """-> (indices_3d, indices_2d), each int64 (num_points + 1,) with the number of correspondences in slot 0, or None
        (projection.py:194-252)"""
        num_points = points.shape[0]
        pix = self.project_frames(points, torch.as_tensor(depth).unsqueeze(0), torch.as_tensor(camera_to_world).unsqueeze(0))[0]
        ind = torch.nonzero(pix >= 0)[:, 0]                       # ascending point index, as the reference's masks keep it
        if ind.numel() == 0:
            return None
        indices_3d = torch.zeros(num_points + 1, dtype=torch.long, device=self.device)
        indices_2d = torch.zeros(num_points + 1, dtype=torch.long, device=self.device)
        indices_3d[0] = indices_2d[0] = ind.numel()
        indices_3d[1:1 + ind.numel()] = ind
        indices_2d[1:1 + ind.numel()] = pix[ind].long()
        return indices_3d, indices_2d

    @torch.no_grad()
    def project(self, label, lin_indices_3d, lin_indices_2d, num_points):
        """projection.py:254-276: scatter the image features of the mapped pixels to their points"""
        num_label_ft = 1 if len(label.shape) == 2 else label.shape[0]
        output = label.new_zeros(num_label_ft, num_points)
        num_ind = int(lin_indices_3d[0])
        if num_ind > 0:
            vals = torch.index_select(label.view(num_label_ft, -1), 1, lin_indices_2d[1:1 + num_ind])
            output.view(num_label_ft, -1)[:, lin_indices_3d[1:1 + num_ind]] = vals
        return output


def save_point_features(path, point_features):
    """the file format of scripts/project_multiview_features.py:201-202, read by lib/dataset.py:408-412"""
    with open(path, "wb") as f:
        pickle.dump(np.array(point_features.detach().cpu().numpy()), f)

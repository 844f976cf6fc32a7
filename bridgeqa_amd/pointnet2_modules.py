"""Set-abstraction and feature-propagation layers -- mirror of the live classes of the
reference's lib/pointnet2/pointnet2_modules.py (PointnetSAModuleVotes :164-277,
PointnetFPModule :361-421).  Same ctor keywords, forward signatures, return tuples and
state-dict names (`mlp_module.layer{i}...` / `mlp.layer{i}...`).
"""
from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import pointnet2_utils
from . import pytorch_utils as pt_utils


class PointnetSAModuleVotes(nn.Module):
    """FPS -> gather centres -> ball-query grouping -> SharedMLP -> max over neighbours.

    forward(xyz (B,N,3), features (B,C,N) | None, inds (B,npoint) i32 | None)
      -> new_xyz (B,npoint,3), new_features (B,mlp[-1],npoint), inds (B,npoint) i32
    """

    def __init__(self, *, mlp: List[int], npoint: int = None, radius: float = None, nsample: int = None,
                 bn: bool = True, use_xyz: bool = True, pooling: str = "max", sigma: float = None,
                 normalize_xyz: bool = False, sample_uniformly: bool = False, ret_unique_cnt: bool = False):
        super().__init__()
        if npoint is None:
            raise NotImplementedError("GroupAll (npoint=None) has no caller in BridgeQA")
        if pooling != "max":
            raise NotImplementedError("only pooling='max' is used by BridgeQA (pointnet2_modules.py:259-262)")
        self.npoint, self.radius, self.nsample = npoint, radius, nsample
        self.pooling, self.use_xyz = pooling, use_xyz
        self.sigma = sigma if sigma is not None else self.radius / 2
        self.normalize_xyz, self.ret_unique_cnt = normalize_xyz, ret_unique_cnt
        self.grouper = pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz, ret_grouped_xyz=True,
                                                     normalize_xyz=normalize_xyz, sample_uniformly=sample_uniformly,
                                                     ret_unique_cnt=ret_unique_cnt)
        mlp_spec = mlp  # NB the reference mutates the caller's list too (:204-206)
        if use_xyz and len(mlp_spec) > 0:
            mlp_spec[0] += 3
        self.mlp_module = pt_utils.SharedMLP(mlp_spec, bn=bn)

    def sample_and_query(self, xyz, inverted=False):
        """the parameter-free part of forward: (inds, new_xyz, group_idx) for coordinates xyz (B, N, 3); inverted: also the
        inverted group index (start, slots) the deterministic grouping gradient gathers over (pointnet2_utils.invert_groups)"""
        with torch.no_grad():
            inds = pointnet2_utils.furthest_point_sample(xyz, self.npoint)
            new_xyz = pointnet2_utils.gather_operation(xyz.transpose(1, 2).contiguous(), inds) \
                .transpose(1, 2).contiguous()
            idx = pointnet2_utils.ball_query(self.grouper.radius, self.grouper.nsample, xyz, new_xyz)
            if inverted:
                return inds, new_xyz, idx, pointnet2_utils.invert_groups(idx, xyz.shape[1])
        return inds, new_xyz, idx

    def forward(self, xyz, features=None, inds=None, geometry=None):
        """geometry: optional (inds, new_xyz, group_idx[, (inv_start, inv_slots)]) from sample_and_query(xyz), computed
        ahead of time"""
        group_idx = inv = None
        if geometry is not None:
            inds, new_xyz, group_idx = geometry[:3]
            inv = geometry[3] if len(geometry) > 3 else None
            assert inds.shape[1] == self.npoint
        else:
            if inds is None:
                inds = pointnet2_utils.furthest_point_sample(xyz, self.npoint)
            else:
                assert inds.shape[1] == self.npoint
            new_xyz = pointnet2_utils.gather_operation(xyz.transpose(1, 2).contiguous(), inds) \
                .transpose(1, 2).contiguous()
        grouped_features, _grouped_xyz = self.grouper(xyz, new_xyz, features, idx=group_idx, inv=inv)  # (B, 3+C, npoint, nsample)
        # max over nsample == F.max_pool2d(kernel=[1, nsample]).squeeze(-1) (pointnet2_modules.py:259-262, 272); a row
        # reduction instead of the generic NCHW pooling kernel.  Tie routing in backward is immaterial.
        if grouped_features.dtype == torch.bfloat16 and (
                pt_utils._rows_view_ok(grouped_features) is not None
                or (grouped_features.is_contiguous(memory_format=torch.channels_last)
                    and not grouped_features.is_contiguous())):
            # NHWC fast path: the last layer's BatchNorm+ReLU kernel also reduces over S; the result stays
            # point-major (B,M,C) and is handed on as a (B,C,M) VIEW so the next level can group it without a transpose
            new_features = self.mlp_module(grouped_features, pool=True).float().transpose(1, 2)
        else:
            new_features = self.mlp_module(grouped_features)
            new_features = new_features.max(dim=3)[0].float()  # module boundary stays fp32 (reference dtype)
        return new_xyz, new_features, inds


class PointnetFPModule(nn.Module):
    """three-NN inverse-distance interpolation of `known_feats` onto `unknown`, concat skip, SharedMLP.

    forward(unknown (B,n,3), known (B,m,3), unknow_feats (B,C1,n), known_feats (B,C2,m)) -> (B,mlp[-1],n)
    """

    def __init__(self, *, mlp: List[int], bn: bool = True):
        super().__init__()
        self.mlp = pt_utils.SharedMLP(mlp, bn=bn)

    def forward(self, unknown, known, unknow_feats, known_feats, nn=None):
        """nn: optional precomputed three_nn(unknown, known) = (dist, idx)"""
        if known is not None:
            dist, idx = nn if nn is not None else pointnet2_utils.three_nn(unknown, known)
            dist_recip = 1.0 / (dist + 1e-8)
            norm = torch.sum(dist_recip, dim=2, keepdim=True)
            weight = dist_recip / norm
            interpolated_feats = pointnet2_utils.three_interpolate(known_feats.contiguous(), idx, weight)
        else:
            interpolated_feats = known_feats.expand(*known_feats.size()[0:2], unknown.size(1))
        if unknow_feats is not None:
            new_features = torch.cat([interpolated_feats, unknow_feats], dim=1)
        else:
            new_features = interpolated_feats
        if (pt_utils.native_rows_ok(new_features) and self.training
                and all(pt_utils._native_layer_ok(layer) for layer in self.mlp) and new_features.shape[1] % 8 == 0):
            # point-major bf16 rows through the native SharedMLP layers (GEMM + BatchNorm statistics in its epilogue);
            # the (B, C, n) result is a transposed view of the rows, fp32 at the module boundary
            B, C, n = new_features.shape
            rows = pt_utils.to_rows(new_features)
            out = self.mlp(rows.view(B, n, 1, C).permute(0, 3, 1, 2))
            return out.squeeze(-1).float()
        return self.mlp(new_features.unsqueeze(-1)).squeeze(-1)

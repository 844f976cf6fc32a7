"""Pointnet2Backbone -- mirror of the reference's models/backbone_module.py:11-131.

4 set-abstraction levels (2048/0.2/64, 1024/0.4/32, 512/0.8/16, 256/1.2/16) + 2 feature
propagation levels; writes sa{1..4}_{xyz,features}, sa1_inds, sa2_inds, fp2_{features,xyz,inds}
into data_dict.  State-dict names: sa{i}.mlp_module.layer{j}.{conv.weight,bn.bn.*}, fp{i}.mlp.layer{j}...
"""
import torch
import torch.nn as nn

from .pointnet2_modules import PointnetFPModule, PointnetSAModuleVotes

# (npoint, radius, nsample) per level -- backbone_module.py:28-66
SA_LEVELS = ((2048, 0.2, 64), (1024, 0.4, 32), (512, 0.8, 16), (256, 1.2, 16))


class Pointnet2Backbone(nn.Module):
    def __init__(self, input_feature_dim=0, width=1, depth=2, seed_feat_dim=256):
        super().__init__()
        self.input_feature_dim = input_feature_dim
        w = width
        chans = (
            [input_feature_dim] + [64 * w] * depth + [128 * w],
            [128 * w] + [128 * w] * depth + [256 * w],
            [256 * w] + [128 * w] * depth + [256 * w],
            [256 * w] + [128 * w] * depth + [256 * w],
        )
        for i, ((npoint, radius, nsample), mlp) in enumerate(zip(SA_LEVELS, chans), start=1):
            setattr(self, "sa%d" % i, PointnetSAModuleVotes(npoint=npoint, radius=radius, nsample=nsample, mlp=mlp,
                                                            use_xyz=True, normalize_xyz=True))
        self.fp1 = PointnetFPModule(mlp=[256 * w + 256 * w, 256 * w, 256 * w])
        self.fp2 = PointnetFPModule(mlp=[256 * w + 256 * w, 256 * w, seed_feat_dim])

    def _break_up_pc(self, pc):
        xyz = pc[..., :3].contiguous()
        if pc.size(-1) <= 3:
            return xyz, None
        from . import fusion_ops
        features = pc[..., 3:].transpose(1, 2)  # (B,C,N) view of the interleaved cloud
        if pc.is_cuda and fusion_ops.POINT_MAJOR[0] and fusion_ops.compute_dtype() == torch.bfloat16:
            return xyz, features  # point-major fast path reads the rows in place: no 340 MB transposing copy
        return xyz, features.contiguous()

    def precompute_geometry(self, point_clouds):
        """Everything in forward() that depends on coordinates only -- FPS picks, sampled centres and ball-query
        groups of the four SA levels, three-NN of the two FP levels -- as a flat dict of tensors.  No parameters
        are involved, so a training loop can compute it for the NEXT batch while the current step is busy elsewhere
        (pipeline.PhasedTrainStep) and hand it back as data_dict["geometry"]; values are identical to what
        forward() would compute itself."""
        from . import pointnet2_utils
        geo = {}
        xyz = point_clouds[..., :3].contiguous()
        level_xyz = {}
        with torch.no_grad():
            for i in (1, 2, 3, 4):
                n_in = xyz.shape[1]
                inds, xyz, idx = getattr(self, "sa%d" % i).sample_and_query(xyz)
                geo["sa%d_inds" % i], geo["sa%d_xyz" % i], geo["sa%d_group_idx" % i] = inds, xyz, idx
                # the inverted group index the deterministic grouping gradient gathers over (csrc/invert.hip): coordinates
                # only as well, so it is sorted here, ahead of time, instead of inside the backward
                inv = pointnet2_utils.invert_groups(idx, n_in)
                if inv is not None:
                    geo["sa%d_inv_start" % i], geo["sa%d_inv_slots" % i] = inv
                level_xyz[i] = xyz
            geo["fp1_dist"], geo["fp1_idx"] = pointnet2_utils.three_nn(level_xyz[3], level_xyz[4])
            geo["fp2_dist"], geo["fp2_idx"] = pointnet2_utils.three_nn(level_xyz[2], level_xyz[3])
        return geo

    def forward(self, data_dict):
        """data_dict["point_clouds"]: (B, N, 3 + input_feature_dim) f32, xyz first; optional data_dict["geometry"]
        from precompute_geometry(point_clouds)."""
        xyz, features = self._break_up_pc(data_dict["point_clouds"])
        geo = data_dict.get("geometry")
        for i in (1, 2, 3, 4):
            g = (geo["sa%d_inds" % i], geo["sa%d_xyz" % i], geo["sa%d_group_idx" % i]) if geo is not None else None
            if g is not None and ("sa%d_inv_start" % i) in geo:
                g = g + ((geo["sa%d_inv_start" % i], geo["sa%d_inv_slots" % i]),)
            xyz, features, inds = getattr(self, "sa%d" % i)(xyz, features, geometry=g)
            if i <= 2:
                data_dict["sa%d_inds" % i] = inds
            data_dict["sa%d_xyz" % i] = xyz
            data_dict["sa%d_features" % i] = features.contiguous()  # reference layout for consumers
        features = self.fp1(data_dict["sa3_xyz"], data_dict["sa4_xyz"], data_dict["sa3_features"],
                            data_dict["sa4_features"], nn=(geo["fp1_dist"], geo["fp1_idx"]) if geo is not None else None)
        features = self.fp2(data_dict["sa2_xyz"], data_dict["sa3_xyz"], data_dict["sa2_features"], features,
                            nn=(geo["fp2_dist"], geo["fp2_idx"]) if geo is not None else None)
        data_dict["fp2_features"] = features
        data_dict["fp2_xyz"] = data_dict["sa2_xyz"]
        num_seed = data_dict["fp2_xyz"].shape[1]
        # seeds are the first num_seed FPS picks of level 1 (FPS prefix property, backbone_module.py:130)
        data_dict["fp2_inds"] = data_dict["sa1_inds"][:, 0:num_seed]
        return data_dict

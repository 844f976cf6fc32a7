"""ProposalModule ("vote-cluster" + proposal head) -- mirror of the reference's
models/proposal_module.py:20-151.

Differences that do not change any value a caller reads:
  * box corners (`bbox_corner`) are decoded ON DEVICE in fp32; the reference round-trips through
    numpy float64 on the host (proposal_module.py:87-108) and thereby synchronises the stream
    every step.  Nothing in the reference consumes bbox_corner on the training path.
  * `ScannetDatasetConfig` (data/scannet/model_util_scannet.py) is ABSENT from the reference tree;
    the few members this module needs are restated in `DatasetConfigLite`.
"""
import numpy as np
import torch
import torch.nn as nn

from .pointnet2_modules import PointnetSAModuleVotes


class DatasetConfigLite(object):
    """The part of VoteNet/ScanRefer's ScannetDatasetConfig used on this path: ScanNet boxes are
    axis-aligned, so class2angle == 0 and size = mean_size_arr[class] + residual."""

    def __init__(self, num_class=18, num_heading_bin=1, num_size_cluster=18, mean_size_arr=None):
        self.num_class, self.num_heading_bin, self.num_size_cluster = num_class, num_heading_bin, num_size_cluster
        if mean_size_arr is None:
            mean_size_arr = np.ones((num_size_cluster, 3), dtype=np.float64)
        self.mean_size_arr = np.asarray(mean_size_arr)


_CORNER_SIGNS = {}   # (device, dtype) -> (8, 3) sign table of box_corners (created once, outside any graph capture: the first
#                       call is an eager warm-up step)


def box_corners(center, size, heading):
    """utils/box_util.py:302-325 get_3d_box_batch, on device.  center (...,3), size (...,3) = (l,w,h),
    heading (...) -> (...,8,3); rotation about the last axis named `roty` in the reference."""
    # corner signs (x: l, y: w, z: h) in the reference's corner order; multiplying by +-1 is exact, so the values equal its
    # cat([l, l, -l, -l, ...]) form bit for bit -- as THREE launches instead of twelve negations and three concatenations
    key = (size.device, size.dtype)
    sg = _CORNER_SIGNS.get(key)
    if sg is None:
        sg = _CORNER_SIGNS[key] = torch.tensor([[1, 1, 1], [1, -1, 1], [-1, -1, 1], [-1, 1, 1], [1, 1, -1], [1, -1, -1],
                                                [-1, -1, -1], [-1, 1, -1]], dtype=size.dtype, device=size.device)
    corners = (size / 2).unsqueeze(-2) * sg            # (..., 8, 3)
    cx, cy, cz = corners[..., 0], corners[..., 1], corners[..., 2]
    c, s = torch.cos(heading).unsqueeze(-1), torch.sin(heading).unsqueeze(-1)
    # corners @ R^T with R = [[c,0,s],[0,1,0],[-s,0,c]]
    x = c * cx + s * cz
    y = cy
    z = -s * cx + c * cz
    return torch.stack([x, y, z], -1) + center.unsqueeze(-2)


class ProposalModule(nn.Module):
    def __init__(self, num_class, num_heading_bin, num_size_cluster, mean_size_arr, num_proposal, sampling,
                 seed_feat_dim=256, proposal_size=128, radius=0.3, nsample=16):
        super().__init__()
        self.num_class, self.num_heading_bin, self.num_size_cluster = num_class, num_heading_bin, num_size_cluster
        self.mean_size_arr = mean_size_arr
        self.num_proposal, self.sampling, self.seed_feat_dim = num_proposal, sampling, seed_feat_dim
        self.votenet_hidden_size = proposal_size
        self.vote_aggregation = PointnetSAModuleVotes(
            npoint=num_proposal, radius=radius, nsample=nsample,
            mlp=[seed_feat_dim, proposal_size, proposal_size, proposal_size], use_xyz=True, normalize_xyz=True)
        out_ch = 2 + 3 + num_heading_bin * 2 + num_size_cluster * 4 + num_class
        self.proposal = nn.Sequential(
            nn.Conv1d(proposal_size, proposal_size, 1, bias=False), nn.BatchNorm1d(proposal_size), nn.ReLU(),
            nn.Conv1d(proposal_size, proposal_size, 1, bias=False), nn.BatchNorm1d(proposal_size), nn.ReLU(),
            nn.Conv1d(proposal_size, out_ch, 1))
        self.register_buffer("_mean_size", torch.from_numpy(np.asarray(mean_size_arr).astype(np.float32)),
                             persistent=False)  # not in the state dict: key set must equal the reference's

    def _proposal_head(self, features):
        """self.proposal (conv-bn-relu x2 + conv, proposal_module.py:44-50); on the bf16 GPU path the two conv + BatchNorm
        + ReLU stages run as native point-major layers and the last convolution as a linear on the rows"""
        from . import pytorch_utils as pt_utils
        p = self.proposal
        if (pt_utils.native_rows_ok(features) and self.training and features.shape[1] % 8 == 0
                and pt_utils.rows_layer_ok(p[0], p[1]) and pt_utils.rows_layer_ok(p[3], p[4])):
            # (both layers' preconditions are checked before either runs: no fallback after a BatchNorm update)
            B, C, K = features.shape
            h = pt_utils.rows_conv_bn_relu(pt_utils.to_rows(features), p[0], p[1])
            h = pt_utils.rows_conv_bn_relu(h, p[3], p[4])
            net = pt_utils.rows_linear_f32(h, p[6].weight.squeeze(-1), p[6].bias)
            return net.view(B, K, -1).transpose(1, 2)
        return p(features)

    def forward(self, xyz, features, data_dict):
        """xyz (B,K,3) votes, features (B,C,K) -> data_dict with the proposal outputs."""
        xyz, features, fps_inds = self.vote_aggregation(xyz, features)
        data_dict["aggregated_vote_xyz"] = xyz
        data_dict["aggregated_vote_features"] = features.permute(0, 2, 1).contiguous()
        data_dict["aggregated_vote_inds"] = fps_inds
        net = self._proposal_head(features)
        return self.decode_scores(net, data_dict, self.num_class, self.num_heading_bin, self.num_size_cluster,
                                  self.mean_size_arr)

    def decode_pred_box(self, data_dict):
        heading_class = torch.argmax(data_dict["heading_scores"], -1)
        size_class = torch.argmax(data_dict["size_scores"], -1)
        size_residual = torch.gather(data_dict["size_residuals"], 2,
                                     size_class[..., None, None].expand(-1, -1, 1, 3)).squeeze(2)
        box_size = self._mean_size.to(size_residual.dtype)[size_class] + size_residual
        heading = torch.zeros_like(heading_class, dtype=box_size.dtype)  # class2angle == 0 for ScanNet
        return box_corners(data_dict["center"].detach(), box_size.detach(), heading * -1)

    def decode_scores(self, net, data_dict, num_class, num_heading_bin, num_size_cluster, mean_size_arr):
        net_t = net.transpose(2, 1).contiguous()  # (B, num_proposal, channels)
        B, K = net_t.shape[0], net_t.shape[1]
        NH, NS = num_heading_bin, num_size_cluster
        o = 0
        objectness_scores = net_t[:, :, o:o + 2]; o += 2
        center = data_dict["aggregated_vote_xyz"] + net_t[:, :, o:o + 3]; o += 3
        heading_scores = net_t[:, :, o:o + NH]; o += NH
        heading_residuals_normalized = net_t[:, :, o:o + NH]; o += NH
        size_scores = net_t[:, :, o:o + NS]; o += NS
        size_residuals_normalized = net_t[:, :, o:o + NS * 3].view(B, K, NS, 3); o += NS * 3
        sem_cls_scores = net_t[:, :, o:]
        data_dict["objectness_scores"] = objectness_scores
        data_dict["center"] = center
        data_dict["heading_scores"] = heading_scores
        data_dict["heading_residuals_normalized"] = heading_residuals_normalized
        data_dict["heading_residuals"] = heading_residuals_normalized * (np.pi / NH)
        data_dict["size_scores"] = size_scores
        data_dict["size_residuals_normalized"] = size_residuals_normalized
        data_dict["size_residuals"] = size_residuals_normalized * self._mean_size.to(net_t.dtype)[None, None]
        data_dict["sem_cls_scores"] = sem_cls_scores
        data_dict["bbox_corner"] = self.decode_pred_box(data_dict)
        data_dict["bbox_feature"] = data_dict["aggregated_vote_features"]
        data_dict["bbox_mask"] = objectness_scores.argmax(-1)
        data_dict["bbox_sems"] = sem_cls_scores.argmax(-1)
        return data_dict

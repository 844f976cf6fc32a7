"""VotingModule -- mirror of the reference's models/voting_module.py:11-60.

Three 1x1 Conv1d (with bias; BatchNorm1d + ReLU after the first two) map each seed feature to
`vote_factor` (xyz offset(3), feature residual(C)) pairs.  State-dict names conv{1,2,3}, bn{1,2}.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import pytorch_utils as pt_utils


class VotingModule(nn.Module):
    def __init__(self, vote_factor, seed_feature_dim):
        super().__init__()
        self.vote_factor = vote_factor
        self.in_dim = seed_feature_dim
        self.out_dim = self.in_dim  # residual features: in == out
        self.conv1 = nn.Conv1d(self.in_dim, self.in_dim, 1)
        self.conv2 = nn.Conv1d(self.in_dim, self.in_dim, 1)
        self.conv3 = nn.Conv1d(self.in_dim, (3 + self.out_dim) * self.vote_factor, 1)
        self.bn1 = nn.BatchNorm1d(self.in_dim)
        self.bn2 = nn.BatchNorm1d(self.in_dim)

    def forward(self, seed_xyz, seed_features):
        """seed_xyz (B,S,3), seed_features (B,C,S) -> vote_xyz (B,S*vf,3), vote_features (B,C,S*vf)"""
        B, S = seed_xyz.shape[0], seed_xyz.shape[1]
        vf, C = self.vote_factor, self.out_dim
        net = None
        if (pt_utils.native_rows_ok(seed_features) and self.training and seed_features.shape[1] % 8 == 0
                and pt_utils.rows_layer_ok(self.conv1, self.bn1) and pt_utils.rows_layer_ok(self.conv2, self.bn2)):
            # point-major rows: conv + BatchNorm + ReLU twice on the native layer (csrc/gemm.hip pwconv + csrc/bn.hip),
            # the last convolution (259 output channels, no BatchNorm) on the same GEMM family with fp32 results.  (Both layers'
            # preconditions are checked before either runs: no fallback after a BatchNorm update.)
            rows = pt_utils.to_rows(seed_features)
            h = pt_utils.rows_conv_bn_relu(rows, self.conv1, self.bn1)
            h = pt_utils.rows_conv_bn_relu(h, self.conv2, self.bn2)
            net = pt_utils.rows_linear_f32(h, self.conv3.weight.squeeze(-1), self.conv3.bias)
            net = net.view(B, S, -1).transpose(1, 2)
        if net is None:
            net = F.relu(self.bn1(self.conv1(seed_features)))
            net = F.relu(self.bn2(self.conv2(net)))
            net = self.conv3(net)  # (B, (3+C)*vf, S); channel = v*(3+C) + [offset(3) | residual(C)]
        net = net.view(B, vf, 3 + C, S)
        offset = net[:, :, 0:3, :].permute(0, 3, 1, 2)  # (B,S,vf,3)
        vote_xyz = (seed_xyz.unsqueeze(2) + offset).reshape(B, S * vf, 3)
        residual = net[:, :, 3:, :]  # (B,vf,C,S)
        vote_features = seed_features.unsqueeze(1) + residual  # (B,vf,C,S)
        # vote index = s*vf + v  (voting_module.py:49-58)
        vote_features = vote_features.permute(0, 2, 3, 1).reshape(B, C, S * vf)
        return vote_xyz.contiguous(), vote_features.contiguous()

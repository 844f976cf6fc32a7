"""huber_loss / nn_distance -- mirror of the reference's utils/nn_distance.py:6-52 (same names, arguments, outputs).

The reference materialises both operands tiled to (B,N,M,C) with .repeat before subtracting (:41-43); broadcasting
gives the same values without the two copies."""
import torch


def huber_loss(error, delta=1.0):
    """0.5 |x|^2 if |x| <= delta else 0.5 delta^2 + delta (|x| - delta)   (nn_distance.py:6-23)"""
    abs_error = torch.abs(error)
    quadratic = torch.clamp(abs_error, max=delta)
    linear = abs_error - quadratic
    return 0.5 * quadratic ** 2 + delta * linear


def nn_distance(pc1, pc2, l1smooth=False, delta=1.0, l1=False):
    """pc1 (B,N,C), pc2 (B,M,C) -> dist1 (B,N), idx1 (B,N) int64, dist2 (B,M), idx2 (B,M)   (nn_distance.py:25-52):
    squared-L2 (default), L1 (l1=True) or Huber (l1smooth=True) distance to the nearest point of the other set."""
    pc_diff = pc1.unsqueeze(2) - pc2.unsqueeze(1)
    if l1smooth:
        pc_dist = torch.sum(huber_loss(pc_diff, delta), dim=-1)
    elif l1:
        pc_dist = torch.sum(torch.abs(pc_diff), dim=-1)
    else:
        pc_dist = torch.sum(pc_diff ** 2, dim=-1)
    dist1, idx1 = torch.min(pc_dist, dim=2)
    dist2, idx2 = torch.min(pc_dist, dim=1)
    return dist1, idx1, dist2, idx2

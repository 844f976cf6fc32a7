"""Autograd operators over the HIP library -- mirror of the reference's
lib/pointnet2/pointnet2_utils.py:51-376 (same names, argument order and autograd contract:
FPS / ball-query outputs non-differentiable, three_nn has no backward, gather / group /
interpolate backward call the `_grad` operator on a contiguous grad_out).

`_ext` is the operator backend (bridgeqa_amd._ext -> libbqhip.so).  Tests may substitute another
module with the same nine functions via `set_backend` (the CPU oracle, for host-logic tests on a
machine without a GPU); the product never does.
"""
import torch
import torch.nn as nn
from torch.autograd import Function

from . import _ext as _hip_ext

_ext = _hip_ext


def set_backend(module):
    """TEST HOOK: swap the operator backend (returns the previous one)."""
    global _ext
    prev, _ext = _ext, module
    return prev


def backend():
    return _ext


def backend_is_hip():
    return _ext is _hip_ext


_BACKGROUND_GEOMETRY = [False]


class background_geometry(object):
    """with background_geometry(): ... -- the ball queries issued inside run on bq_ball_query_background's grid (about
    one workgroup per CU, same results).  For geometry computed AHEAD of time on a second stream (pipeline.PhasedTrainStep's
    prefetch of the next batch's indices under the fusion phase): the full grid holds every wave slot of the chip and the
    other stream's latency-bound kernels queue behind it.  No effect on other backends (the CPU oracle in tests)."""

    def __enter__(self):
        self.prev, _BACKGROUND_GEOMETRY[0] = _BACKGROUND_GEOMETRY[0], backend_is_hip()
        return self

    def __exit__(self, *a):
        _BACKGROUND_GEOMETRY[0] = self.prev


class FurthestPointSampling(Function):
    """xyz (B,N,3) f32, npoint -> (B,npoint) i32   [pointnet2_utils.py:51-80]"""

    @staticmethod
    def forward(ctx, xyz, npoint):
        inds = _ext.furthest_point_sampling(xyz, npoint)
        ctx.mark_non_differentiable(inds)
        return inds

    @staticmethod
    def backward(ctx, grad=None):
        return None, None


furthest_point_sample = FurthestPointSampling.apply


class GatherOperation(Function):
    """features (B,C,N), idx (B,npoint) -> (B,C,npoint)   [pointnet2_utils.py:83-117]"""

    @staticmethod
    def forward(ctx, features, idx):
        ctx.save_for_backward(idx)
        ctx.n = features.size(2)
        return _ext.gather_points(features, idx)

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        return _ext.gather_points_grad(grad_out.contiguous(), idx, ctx.n), None


gather_operation = GatherOperation.apply


class ThreeNN(Function):
    """unknown (B,n,3), known (B,m,3) -> (dist (B,n,3) = sqrt(d2), idx (B,n,3))   [:120-149]"""

    @staticmethod
    def forward(ctx, unknown, known):
        if hasattr(_ext, "three_nn_dist"):
            dist, idx = _ext.three_nn_dist(unknown, known)
        else:
            dist2, idx = _ext.three_nn(unknown, known)
            dist = torch.sqrt(dist2)
        ctx.mark_non_differentiable(idx)
        return dist, idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    """features (B,c,m), idx (B,n,3), weight (B,n,3) -> (B,c,n)   [:152-206]"""

    @staticmethod
    def forward(ctx, features, idx, weight):
        ctx.save_for_backward(idx, weight)
        ctx.m = features.size(2)
        return _ext.three_interpolate(features, idx, weight)

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight = ctx.saved_tensors
        if _det_scatter(grad_out):
            # gather over the inverted index: one sum per known point, terms in ascending order (no fp32 atomics)
            inv = _ext.invert_index(idx, ctx.m)
            return _ext.three_interpolate_grad_gather(grad_out.contiguous().float(), inv, weight.contiguous(), ctx.m), None, None
        return _ext.three_interpolate_grad(grad_out.contiguous(), idx, weight, ctx.m), None, None


three_interpolate = ThreeInterpolate.apply


class GroupingOperation(Function):
    """features (B,C,N), idx (B,npoint,nsample) -> (B,C,npoint,nsample)   [:209-257]"""

    @staticmethod
    def forward(ctx, features, idx):
        ctx.save_for_backward(idx)
        ctx.n = features.size(2)
        return _ext.group_points(features, idx)

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        return _ext.group_points_grad(grad_out.contiguous(), idx, ctx.n), None


grouping_operation = GroupingOperation.apply


class BallQuery(Function):
    """(radius, nsample, xyz (B,N,3), new_xyz (B,npoint,3)) -> (B,npoint,nsample) i32   [:260-291]
    NB the operator takes (new_xyz, xyz, radius, nsample)."""

    @staticmethod
    def forward(ctx, radius, nsample, xyz, new_xyz):
        if _BACKGROUND_GEOMETRY[0]:   # (inside background_geometry(): the HIP backend's gentle grid, same results)
            inds = _ext.ball_query(new_xyz, xyz, radius, nsample, background=True)
        else:
            inds = _ext.ball_query(new_xyz, xyz, radius, nsample)
        ctx.mark_non_differentiable(inds)
        return inds

    @staticmethod
    def backward(ctx, grad=None):
        return None, None, None, None


ball_query = BallQuery.apply


def invert_groups(idx, n):
    """inverted group index of ball_query's idx (B, M, S) over n points for the deterministic grouping gradient
    (csrc/invert.hip): (start, slots), or None where the gather path does not apply (CPU oracle backend, switch off)"""
    if not (idx.is_cuda and _ext is _hip_ext and getattr(_ext, "DETERMINISTIC_SCATTER", [False])[0]):
        return None
    return _ext.invert_index(idx, n)


def _det_scatter(t):
    return t.is_cuda and _ext is _hip_ext and getattr(_ext, "DETERMINISTIC_SCATTER", [False])[0]


class _GroupConcat(Function):
    """Fused tail of QueryAndGroup (pointnet2_utils.py:348-359): one kernel writes
    cat([(xyz[idx]-centre)/radius, features[idx]], dim=1) instead of 2 gathers + sub + div + cat."""

    @staticmethod
    def forward(ctx, xyz, new_xyz, features, idx, radius, normalize, out_dtype=torch.float32):
        ctx.save_for_backward(idx)
        ctx.n, ctx.radius, ctx.normalize = xyz.size(1), radius, normalize
        ctx.has_features = features is not None
        if out_dtype != torch.float32:
            return _ext.group_concat(xyz, new_xyz, features, idx, radius, normalize, out_dtype)
        return _ext.group_concat(xyz, new_xyz, features, idx, radius, normalize)

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        need = ctx.needs_input_grad
        gf, gx, gn = _ext.group_concat_grad(grad_out.contiguous(), idx, ctx.n, ctx.radius, ctx.normalize,
                                            ctx.has_features and need[2], need[0], need[1])
        return gx, gn, gf, None, None, None, None


class _GroupConcatPM(Function):
    """Point-major grouping (csrc/pn2_ops.hip group_concat_pm_kernel): feats_pm (B,N,C) rows -> (B,M,S,3+C)."""

    @staticmethod
    def forward(ctx, xyz, new_xyz, feats_pm, idx, radius, normalize, out_dtype, pad_to=1, inv=None):
        ctx.save_for_backward(idx)
        ctx.n, ctx.radius, ctx.normalize = xyz.size(1), radius, normalize
        ctx.has_features = feats_pm is not None
        ctx.inv = inv    # (start, slots) of invert_groups(idx, N) when the caller has it (the geometry prefetch), else None
        return _ext.group_concat_pm(xyz, new_xyz, feats_pm, idx, radius, normalize, out_dtype, pad_to)

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        need = ctx.needs_input_grad
        want_f = ctx.has_features and need[2]
        if _det_scatter(grad_out) and (want_f or need[0] or need[1]):
            # GATHERS over the inverted group index: one sum per point in ascending position order, bitwise reproducible
            # (the vote-aggregation level differentiates through its coordinates too: predicted votes)
            inv = ctx.inv if ctx.inv is not None else _ext.invert_index(idx, ctx.n)
            res = _ext.group_concat_pm_grad_gather(grad_out, inv, ctx.n, ctx.radius, ctx.normalize, want_f, need[0], need[1])
            if res is not None:
                gf, gx, gn = res
                return gx, gn, gf, None, None, None, None, None, None
        gf, gx, gn = _ext.group_concat_pm_grad(grad_out, idx, ctx.n, ctx.radius, ctx.normalize, want_f, need[0], need[1])
        return gx, gn, gf, None, None, None, None, None, None


def point_major(features):
    """(B,C,N) features -> (B,N,C) view with contiguous rows (a free view when `features` already is a transposed
    point-major tensor, one transposing copy otherwise)"""
    if features is None:
        return None
    pm = features.transpose(1, 2)
    return pm if pm.stride(2) == 1 else pm.contiguous()


class QueryAndGroup(nn.Module):
    """Ball query + neighbourhood gather   [pointnet2_utils.py:294-376].

    Only the live configuration of the reference is supported (sample_uniformly=False,
    ret_unique_cnt=False; SURVEY.md §2.1 "live vs dead code").  Returns new_features
    (B, 3+C, npoint, nsample) and, with ret_grouped_xyz, the centred/normalised grouped_xyz
    (B, 3, npoint, nsample) -- a view of the first three channels, as the reference's in-place
    ops make it the same values.
    """

    def __init__(self, radius, nsample, use_xyz=True, ret_grouped_xyz=False, normalize_xyz=False,
                 sample_uniformly=False, ret_unique_cnt=False):
        super().__init__()
        if sample_uniformly or ret_unique_cnt:
            raise NotImplementedError("sample_uniformly / ret_unique_cnt have no caller in BridgeQA")
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz
        self.ret_grouped_xyz = ret_grouped_xyz
        self.normalize_xyz = normalize_xyz

    def forward(self, xyz, new_xyz, features=None, idx=None, inv=None):
        """idx: optional precomputed ball_query(radius, nsample, xyz, new_xyz) (geometry prefetch: the indices depend
        on coordinates only, not on parameters -- Pointnet2Backbone.precompute_geometry); inv: its inverted form
        (invert_groups) when that was precomputed too"""
        if idx is None:
            idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        from . import fusion_ops
        if (xyz.is_cuda and _ext is _hip_ext and fusion_ops.POINT_MAJOR[0] and self.use_xyz
                and fusion_ops.compute_dtype() == torch.bfloat16):
            # rows padded to a multiple of 8 elements (16 bytes): the SharedMLP's first GEMM reads them in place
            out = _GroupConcatPM.apply(xyz, new_xyz, point_major(features), idx, self.radius, self.normalize_xyz,
                                       torch.bfloat16, 8, inv)
            new_features = out.permute(0, 3, 1, 2)  # logical (B,3+C,M,S), physically NHWC
            return (new_features, new_features[:, :3]) if self.ret_grouped_xyz else new_features
        fused = hasattr(_ext, "group_concat") and self.nsample % 4 == 0
        if fused and (self.use_xyz or features is None):
            from . import fusion_ops
            # bf16 grouped tensor + bf16 SharedMLP GEMMs exist (fusion_ops.SHAREDMLP_BF16) but are OFF by default:
            # hipBLASLt's batched GEMM at Cout x Cin <= 256 x 259 loses to MIOpen's fp32 convolution (tools/time_sa.py)
            dt = fusion_ops.compute_dtype() if (xyz.is_cuda and _ext is _hip_ext and fusion_ops.SHAREDMLP_BF16[0]) \
                else torch.float32
            new_features = _GroupConcat.apply(xyz, new_xyz, features, idx, self.radius, self.normalize_xyz, dt)
            grouped_xyz = new_features[:, :3]
        else:
            grouped_xyz = grouping_operation(xyz.transpose(1, 2).contiguous(), idx)
            grouped_xyz = grouped_xyz - new_xyz.transpose(1, 2).unsqueeze(-1)
            if self.normalize_xyz:
                grouped_xyz = grouped_xyz / self.radius
            if features is not None:
                grouped_features = grouping_operation(features, idx)
                new_features = torch.cat([grouped_xyz, grouped_features], dim=1) if self.use_xyz else grouped_features
            else:
                assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
                new_features = grouped_xyz
        if self.ret_grouped_xyz:
            return new_features, grouped_xyz
        return new_features

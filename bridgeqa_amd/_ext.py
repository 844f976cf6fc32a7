"""`pointnet2._ext` for MI355X: the reference's nine-function operator surface
(lib/pointnet2/_ext_src/src/bindings.cpp:6-19) bound to libbqhip.so through ctypes.

What the reference's C++ wrappers do on the host is done here: dtype / contiguity / device checks
(include/utils.h:5-25), output allocation (zero-filled where the kernel accumulates), and turning
a non-zero status into a Python exception (the reference prints and exit(-1)s instead).
There is NO CPU path: like the reference ("CPU not supported", sampling.cpp:34) a host tensor
raises, and a missing library raises at import time.
"""
import ctypes
import os

import torch  # must be imported first: libbqhip.so resolves libamdhip64.so.7 to torch's copy

# (BQHIP_LIB: another build of the same sources, e.g. with a measurement macro set -- A/B runs in one gpurun call)
_LIB_PATH = os.environ.get("BQHIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libbqhip.so")

if not os.path.exists(_LIB_PATH):
    raise ImportError(
        "bridgeqa_amd: %s is missing -- build it with `python -m bridgeqa_amd.build` "
        "(hipcc --offload-arch=gfx950). There is no CPU fallback." % _LIB_PATH)

_lib = ctypes.CDLL(_LIB_PATH)
_lib.bq_last_error.restype = ctypes.c_char_p
_lib.bq_abi_version.restype = ctypes.c_int
ABI_VERSION = 6   # = BQHIP_ABI_VERSION of include/bqhip.h
if _lib.bq_abi_version() != ABI_VERSION:
    raise ImportError("bridgeqa_amd: libbqhip.so ABI %d != %d (stale library: python -m bridgeqa_amd.build --force)"
                      % (_lib.bq_abi_version(), ABI_VERSION))

_vp, _i, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
_SIGS = {
    "bq_opt_n_threads": [_i],
    "bq_furthest_point_sampling": [_vp, _vp, ctypes.c_size_t, _vp, _i, _i, _i, _vp],
    "bq_furthest_point_sampling_bruteforce": [_vp, _vp, _vp, _i, _i, _i, _vp],
    "bq_gather_points": [_vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "bq_gather_points_grad": [_vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "bq_ball_query": [_vp, _vp, _vp, _i, _i, _i, _f, _i, _vp],
    "bq_ball_query_background": [_vp, _vp, _vp, _i, _i, _i, _f, _i, _vp],
    "bq_group_points": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "bq_group_points_grad": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "bq_three_nn": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "bq_three_nn_dist": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "bq_three_interpolate": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "bq_three_interpolate_grad": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "bq_group_concat": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp],
    "bq_group_concat_grad": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp],
    "bq_group_concat_bf16": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp],
    "bq_group_concat_grad_bf16": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp],
}
for _name, _args in _SIGS.items():
    getattr(_lib, _name).argtypes = _args
    getattr(_lib, _name).restype = ctypes.c_int
_l = ctypes.c_long
_u = ctypes.c_uint
_lib.bq_attn_fwd.argtypes = [_vp] * 6 + [_i] * 5 + [_l] * 9 + [_f, _f, _u, _vp, _i, _vp]
_lib.bq_attn_fwd.restype = ctypes.c_int
_lib.bq_attn_bwd.argtypes = [_vp] * 11 + [_i] * 5 + [_l] * 9 + [_f, _f, _u, _vp, _i, _vp]
_lib.bq_attn_bwd.restype = ctypes.c_int
_lib.bq_attn_set_persistent.argtypes = [_i]
_lib.bq_attn_set_persistent.restype = ctypes.c_int
_lib.bq_attn_probs.argtypes = [_vp] * 5 + [_i] * 5 + [_l] * 6 + [_f, _f, _u, _vp, _i, _vp]
_lib.bq_attn_probs.restype = ctypes.c_int
_lib.bq_drop_add_ln_fwd.argtypes = [_vp] * 9 + [_i, _i, _f, _f, _f, _i, _u, _vp, _vp]
_lib.bq_drop_add_ln_fwd.restype = ctypes.c_int
_lib.bq_drop_add_ln_bwd.argtypes = [_vp] * 10 + [_i, _i, _f, _f, _f, _i, _u, _vp, _vp]
_lib.bq_drop_add_ln_bwd.restype = ctypes.c_int
_lib.bq_drop_add_ln_bwd_sum.argtypes = [_vp] * 9 + [_i, _i, _f, _f, _i, _u, _vp, _vp]
_lib.bq_drop_add_ln_bwd_sum.restype = ctypes.c_int
_lib.bq_twin_drop_add_ln_fwd.argtypes = [_vp] * 10 + [_i, _i, _f, _f, _u, _vp, _vp]
_lib.bq_twin_drop_add_ln_fwd.restype = ctypes.c_int
_lib.bq_twin_drop_add_ln_bwd.argtypes = [_vp] * 10 + [_i, _i, _f, _f, _u, _vp, _vp]
_lib.bq_twin_drop_add_ln_bwd.restype = ctypes.c_int
_lib.bq_fps_workspace_bytes.argtypes = [_i, _i]
_lib.bq_fps_workspace_bytes.restype = ctypes.c_size_t


def library_path():
    return _LIB_PATH


def _check(status, what):
    if status != 0:
        raise RuntimeError("%s failed (status %d): %s" % (what, status, _lib.bq_last_error().decode()))


def _req(t, dtype, name):
    if not t.is_cuda:
        raise RuntimeError("%s: CPU not supported (bridgeqa_amd has no CPU path)" % name)
    if not t.is_contiguous():
        raise RuntimeError("%s must be a contiguous tensor" % name)
    if t.dtype != dtype:
        raise RuntimeError("%s must be a %s tensor" % (name, "float" if dtype == torch.float32 else "int"))


def _same_device(*ts):
    d = ts[0].device
    for t in ts[1:]:
        if t.device != d:
            raise RuntimeError("all tensors must be on the same device")


def _p(t):
    return t.data_ptr() if t is not None else None


def _stream():
    return torch.cuda.current_stream().cuda_stream


def opt_n_threads(n):
    return _lib.bq_opt_n_threads(int(n))


def furthest_point_sampling(points, nsamples):
    _req(points, torch.float32, "points")
    B, N, _ = points.shape
    with torch.cuda.device(points.device):
        out = torch.zeros(B, nsamples, dtype=torch.int32, device=points.device)
        nbytes = _lib.bq_fps_workspace_bytes(B, N)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=points.device) if nbytes else None
        _check(_lib.bq_furthest_point_sampling(_p(points), _p(ws), nbytes, _p(out), B, N, int(nsamples), _stream()),
               "furthest_point_sampling")
    return out


def furthest_point_sampling_bruteforce(points, nsamples):
    """Unpruned kernels (A/B timing, independent cross-check of the bucketed kernel)."""
    _req(points, torch.float32, "points")
    B, N, _ = points.shape
    with torch.cuda.device(points.device):
        out = torch.zeros(B, nsamples, dtype=torch.int32, device=points.device)
        tmp = torch.empty(B, N, dtype=torch.float32, device=points.device) if N > 24576 else None
        _check(_lib.bq_furthest_point_sampling_bruteforce(_p(points), _p(tmp), _p(out), B, N, int(nsamples),
                                                          _stream()), "furthest_point_sampling_bruteforce")
    return out


def gather_points(points, idx):
    _req(points, torch.float32, "points"); _req(idx, torch.int32, "idx"); _same_device(points, idx)
    B, C, N = points.shape
    M = idx.shape[1]
    with torch.cuda.device(points.device):
        out = torch.empty(B, C, M, dtype=torch.float32, device=points.device)
        _check(_lib.bq_gather_points(_p(points), _p(idx), _p(out), B, C, N, M, _stream()), "gather_points")
    return out


def gather_points_grad(grad_out, idx, n):
    _req(grad_out, torch.float32, "grad_out"); _req(idx, torch.int32, "idx"); _same_device(grad_out, idx)
    B, C, M = grad_out.shape
    with torch.cuda.device(grad_out.device):
        out = torch.zeros(B, C, n, dtype=torch.float32, device=grad_out.device)
        _check(_lib.bq_gather_points_grad(_p(grad_out), _p(idx), _p(out), B, C, int(n), M, _stream()),
               "gather_points_grad")
    return out


def ball_query(new_xyz, xyz, radius, nsample, background=False):
    """background: bq_ball_query_background (same results on about one workgroup per CU: for a query issued ahead of time on
    a second stream, include/bqhip.h)"""
    _req(new_xyz, torch.float32, "new_xyz"); _req(xyz, torch.float32, "xyz"); _same_device(new_xyz, xyz)
    B, M, _ = new_xyz.shape
    N = xyz.shape[1]
    with torch.cuda.device(xyz.device):
        idx = torch.empty(B, M, nsample, dtype=torch.int32, device=xyz.device)
        if N >= BALL_QUERY_GRID_MIN_N[0] and B > 0 and M > 0 and nsample > 0 and radius > 0:
            # large scenes: the same indices through a uniform grid (csrc/ball_query_grid.hip), ~400x fewer distance tests
            nbytes = int(_lib.bq_ball_query_grid_workspace_bytes(B, N))
            ws = torch.empty(nbytes, dtype=torch.uint8, device=xyz.device)
            _check(_lib.bq_ball_query_grid(_p(new_xyz), _p(xyz), _p(idx), B, N, M, float(radius), int(nsample), _p(ws), nbytes,
                                           _stream()), "ball_query_grid")
            return idx
        fn = _lib.bq_ball_query_background if background else _lib.bq_ball_query
        _check(fn(_p(new_xyz), _p(xyz), _p(idx), B, N, M, float(radius), int(nsample), _stream()), "ball_query")
    return idx


BALL_QUERY_GRID_MIN_N = [8192]   # scenes from this size on go through the grid (tests / tools set it to compare the two paths)
_lib.bq_ball_query_grid_workspace_bytes.argtypes = [_i, _i]
_lib.bq_ball_query_grid_workspace_bytes.restype = ctypes.c_size_t
_lib.bq_ball_query_grid.argtypes = [_vp, _vp, _vp, _i, _i, _i, _f, _i, _vp, ctypes.c_size_t, _vp]
_lib.bq_ball_query_grid.restype = ctypes.c_int
_lib.bq_ball_query_grid_build_mode.argtypes = [_i]
_lib.bq_ball_query_grid_build_mode.restype = ctypes.c_int


def ball_query_grid_build_mode(multi):
    """1: the multi-workgroup grid build (default), 0: one workgroup per scene; returns the previous mode (include/bqhip.h)"""
    return int(_lib.bq_ball_query_grid_build_mode(int(multi)))


def group_points(points, idx):
    _req(points, torch.float32, "points"); _req(idx, torch.int32, "idx"); _same_device(points, idx)
    B, C, N = points.shape
    _, M, S = idx.shape
    with torch.cuda.device(points.device):
        out = torch.empty(B, C, M, S, dtype=torch.float32, device=points.device)
        _check(_lib.bq_group_points(_p(points), _p(idx), _p(out), B, C, N, M, S, _stream()), "group_points")
    return out


def group_points_grad(grad_out, idx, n):
    _req(grad_out, torch.float32, "grad_out"); _req(idx, torch.int32, "idx"); _same_device(grad_out, idx)
    B, C, M, S = grad_out.shape
    with torch.cuda.device(grad_out.device):
        out = torch.zeros(B, C, n, dtype=torch.float32, device=grad_out.device)
        _check(_lib.bq_group_points_grad(_p(grad_out), _p(idx), _p(out), B, C, int(n), M, S, _stream()),
               "group_points_grad")
    return out


def three_nn(unknowns, knows):
    _req(unknowns, torch.float32, "unknowns"); _req(knows, torch.float32, "knows"); _same_device(unknowns, knows)
    B, n, _ = unknowns.shape
    m = knows.shape[1]
    with torch.cuda.device(unknowns.device):
        dist2 = torch.empty(B, n, 3, dtype=torch.float32, device=unknowns.device)
        idx = torch.empty(B, n, 3, dtype=torch.int32, device=unknowns.device)
        _check(_lib.bq_three_nn(_p(unknowns), _p(knows), _p(dist2), _p(idx), B, n, m, _stream()), "three_nn")
    return [dist2, idx]


def three_nn_dist(unknowns, knows):
    """three_nn + sqrt in one kernel (IEEE-correct sqrt, unlike the device sqrt torch uses)."""
    _req(unknowns, torch.float32, "unknowns"); _req(knows, torch.float32, "knows"); _same_device(unknowns, knows)
    B, n, _ = unknowns.shape
    m = knows.shape[1]
    with torch.cuda.device(unknowns.device):
        dist = torch.empty(B, n, 3, dtype=torch.float32, device=unknowns.device)
        idx = torch.empty(B, n, 3, dtype=torch.int32, device=unknowns.device)
        _check(_lib.bq_three_nn_dist(_p(unknowns), _p(knows), _p(dist), _p(idx), B, n, m, _stream()), "three_nn_dist")
    return [dist, idx]


def three_interpolate(points, idx, weight):
    _req(points, torch.float32, "points"); _req(idx, torch.int32, "idx"); _req(weight, torch.float32, "weight")
    _same_device(points, idx, weight)
    B, C, m = points.shape
    n = idx.shape[1]
    with torch.cuda.device(points.device):
        out = torch.empty(B, C, n, dtype=torch.float32, device=points.device)
        _check(_lib.bq_three_interpolate(_p(points), _p(idx), _p(weight), _p(out), B, C, m, n, _stream()),
               "three_interpolate")
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    _req(grad_out, torch.float32, "grad_out"); _req(idx, torch.int32, "idx"); _req(weight, torch.float32, "weight")
    _same_device(grad_out, idx, weight)
    B, C, n = grad_out.shape
    with torch.cuda.device(grad_out.device):
        out = torch.zeros(B, C, m, dtype=torch.float32, device=grad_out.device)
        _check(_lib.bq_three_interpolate_grad(_p(grad_out), _p(idx), _p(weight), _p(out), B, C, n, int(m), _stream()),
               "three_interpolate_grad")
    return out


# ---- fused forms (include/bqhip.h) ---------------------------------------------------------------
def group_concat(xyz, new_xyz, features, idx, radius, normalize, out_dtype=torch.float32):
    _req(xyz, torch.float32, "xyz"); _req(new_xyz, torch.float32, "new_xyz"); _req(idx, torch.int32, "idx")
    if features is not None:
        _req(features, torch.float32, "features")
    B, N, _ = xyz.shape
    _, M, S = idx.shape
    C = features.shape[1] if features is not None else 0
    with torch.cuda.device(xyz.device):
        out = torch.empty(B, C + 3, M, S, dtype=out_dtype, device=xyz.device)
        fn = _lib.bq_group_concat_bf16 if out_dtype == torch.bfloat16 else _lib.bq_group_concat
        _check(fn(_p(xyz), _p(new_xyz), _p(features), _p(idx), _p(out), B, C, N, M, S,
                  float(radius), int(bool(normalize)), _stream()), "group_concat")
    return out


def group_concat_grad(grad_out, idx, n, radius, normalize, need_features, need_xyz, need_new_xyz):
    _req(grad_out, grad_out.dtype if grad_out.dtype == torch.bfloat16 else torch.float32, "grad_out")
    _req(idx, torch.int32, "idx")
    B, CT, M, S = grad_out.shape
    C = CT - 3
    dev = grad_out.device
    with torch.cuda.device(dev):
        gf = torch.zeros(B, C, n, dtype=torch.float32, device=dev) if (need_features and C > 0) else None
        gx = torch.zeros(B, n, 3, dtype=torch.float32, device=dev) if need_xyz else None
        gn = torch.zeros(B, M, 3, dtype=torch.float32, device=dev) if need_new_xyz else None
        fn = _lib.bq_group_concat_grad_bf16 if grad_out.dtype == torch.bfloat16 else _lib.bq_group_concat_grad
        _check(fn(_p(grad_out), _p(idx), _p(gf), _p(gx), _p(gn), B, C, int(n), M, S,
                  float(radius), int(bool(normalize)), _stream()), "group_concat_grad")
    return gf, gx, gn


_lib.bq_group_concat_pm.argtypes = [_vp, _vp, _vp, ctypes.c_long, ctypes.c_long, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _i, _vp]
_lib.bq_group_concat_pm.restype = ctypes.c_int
_lib.bq_group_concat_pm_grad.argtypes = [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _i, _vp]
_lib.bq_group_concat_pm_grad.restype = ctypes.c_int


def group_concat_pm(xyz, new_xyz, feats_pm, idx, radius, normalize, out_dtype, pad_to=1):
    """Point-major grouping: feats_pm (B,N,C) f32 view with contiguous rows (any batch / row stride) or None ->
    (B, M, S, 3+C) in out_dtype (f32 / bf16).  pad_to > 1: the rows are allocated with a stride of 3+C rounded up to a
    multiple of pad_to (padding zeroed) and the (B, M, S, 3+C) result is a view of them."""
    _req(xyz, torch.float32, "xyz"); _req(new_xyz, torch.float32, "new_xyz"); _req(idx, torch.int32, "idx")
    B, N, _ = xyz.shape
    _, M, S = idx.shape
    C = 0
    fbs = frs = 0
    if feats_pm is not None:
        if not feats_pm.is_cuda or feats_pm.dtype != torch.float32 or feats_pm.stride(2) != 1:
            raise RuntimeError("feats_pm must be a CUDA float tensor (B,N,C) with contiguous rows")
        C, fbs, frs = feats_pm.shape[2], feats_pm.stride(0), feats_pm.stride(1)
    ld = (C + 3 + pad_to - 1) // pad_to * pad_to
    with torch.cuda.device(xyz.device):
        out = torch.empty(B, M, S, ld, dtype=out_dtype, device=xyz.device)
        _check(_lib.bq_group_concat_pm(_p(xyz), _p(new_xyz), _p(feats_pm), fbs, frs, _p(idx), _p(out),
                                       int(out_dtype == torch.bfloat16), B, C, N, M, S, float(radius),
                                       int(bool(normalize)), ld, _stream()), "group_concat_pm")
    return out[..., :C + 3] if ld != C + 3 else out


def group_concat_pm_grad(grad_out, idx, n, radius, normalize, need_features, need_xyz, need_new_xyz):
    """grad_out (B, M, S, 3+C): contiguous, or a view of rows with a larger uniform stride (the padded rows the
    SharedMLP's input gradient is written in)"""
    B, M, S, CT = grad_out.shape
    C = CT - 3
    dev = grad_out.device
    ld = grad_out.stride(2)
    if grad_out.stride(3) != 1 or grad_out.stride(1) != S * ld or grad_out.stride(0) != M * S * ld or ld < CT:
        grad_out = grad_out.contiguous()
        ld = CT
    with torch.cuda.device(dev):
        gf = torch.zeros(B, n, C, dtype=torch.float32, device=dev) if (need_features and C > 0) else None
        gx = torch.zeros(B, n, 3, dtype=torch.float32, device=dev) if need_xyz else None
        gn = torch.zeros(B, M, 3, dtype=torch.float32, device=dev) if need_new_xyz else None
        _check(_lib.bq_group_concat_pm_grad(_p(grad_out), int(grad_out.dtype == torch.bfloat16), _p(idx), _p(gf),
                                            _p(gx), _p(gn), B, C, int(n), M, S, float(radius),
                                            int(bool(normalize)), ld, _stream()), "group_concat_pm_grad")
    return gf, gx, gn


# ---- fused attention (csrc/attn.hip) ---------------------------------------------------------------
def _bhd_strides(t):
    """(B, L, H, 64) view with a contiguous last dim -> element strides (batch, token, head)."""
    if t.dtype != torch.bfloat16 or t.stride(3) != 1 or t.shape[3] != 64:
        raise RuntimeError("attention operands must be bf16 (B, L, H, 64) views with a contiguous head dim")
    return t.stride(0), t.stride(1), t.stride(2)


LOG2E = 1.4426950408889634


def _pad64(n):
    return (n + 63) // 64 * 64


def key_mask_log2(mask, B, Lk):
    """additive key mask broadcastable to (B,1,1,Lk) -> f32 (B, Lkp), multiplied by log2(e), zero in the padding"""
    if mask is None:
        return None
    m = torch.zeros(B, _pad64(Lk), dtype=torch.float32, device=mask.device)
    m[:, :Lk] = mask.reshape(-1, Lk).expand(B, Lk).float() * LOG2E
    return m


def attn_fwd(q, k, v, scale, mask_log2=None, p_drop=0.0, seed=0, seed_tensor=None, causal=False, out=None):
    """softmax(q k^T * scale + mask) v without materialising the scores.  q: bf16 (B, Lq, H, 64) view, k / v:
    (B, Lk, H, 64) views with equal strides; mask_log2 from key_mask_log2.  Returns out (B, Lq, H, 64) bf16 contiguous
    and lse (B, H, Lq) f32 (log2 domain).  No transposed copies: the kernels transpose out of LDS."""
    for t, n in ((q, "q"), (k, "k"), (v, "v")):
        if not t.is_cuda:
            raise RuntimeError("%s: CPU not supported" % n)
    if k.stride() != v.stride():
        raise RuntimeError("attn_fwd: k and v must have equal strides")
    B, Lq, H, D = q.shape
    Lk = k.shape[1]
    Lkp = _pad64(Lk)
    with torch.cuda.device(q.device):
        if out is None:
            out = torch.empty(B, Lq, H, D, dtype=torch.bfloat16, device=q.device)
        elif tuple(out.shape) != (B, Lq, H, D) or out.dtype != torch.bfloat16 or not out.is_contiguous():
            raise RuntimeError("attn_fwd: out must be a contiguous bf16 (B, Lq, H, 64) tensor")
        lse = torch.empty(B, H, Lq, dtype=torch.float32, device=q.device)
        qs, ks, os_ = _bhd_strides(q), _bhd_strides(k), _bhd_strides(out)
        _check(_lib.bq_attn_fwd(_p(q), _p(k), _p(v), _p(out), _p(lse), _p(mask_log2), B, H, Lq, Lk, Lkp, *qs, *ks,
                                *os_, float(scale), float(p_drop), int(seed) & 0xFFFFFFFF, _p(seed_tensor),
                                int(bool(causal)), _stream()), "attn_fwd")
    return out, lse


def attn_probs(q, k, lse, scale, mask_log2=None, p_drop=0.0, seed=0, seed_tensor=None, causal=False):
    """the softmax probabilities of the attn_fwd call that produced `lse`: f32 (B, H, Lq, Lk).  p_drop = 0 (default): the
    map BEFORE dropout, which is what the reference returns (med.py:202,223); with the forward's p_drop / seed: the dropped
    map the context was computed from"""
    B, Lq, H, D = q.shape
    Lk = k.shape[1]
    with torch.cuda.device(q.device):
        P = torch.empty(B, H, Lq, Lk, dtype=torch.float32, device=q.device)
        qs, ks = _bhd_strides(q), _bhd_strides(k)
        _check(_lib.bq_attn_probs(_p(q), _p(k), _p(lse), _p(mask_log2), _p(P), B, H, Lq, Lk, _pad64(Lk), *qs, *ks,
                                  float(scale), float(p_drop), int(seed) & 0xFFFFFFFF, _p(seed_tensor), int(bool(causal)),
                                  _stream()), "attn_probs")
    return P


def attn_set_persistent(mask):
    """which passes of the unmasked attention use the resident-grid kernels (bit 0 forward, 1 dQ, 2 dK/dV); returns the
    previous mask (include/bqhip_fusion.h: bq_attn_set_persistent)"""
    return int(_lib.bq_attn_set_persistent(int(mask)))


def attn_bwd(q, k, v, out, lse, grad_out, scale, dq, dk, dv, mask_log2=None, p_drop=0.0, seed=0, seed_tensor=None,
             causal=False):
    """Backward of attn_fwd (same mask / dropout arguments).  dq, dk, dv: preallocated bf16 views (dq strided like
    q, dk/dv like k; k and v with equal strides)."""
    B, Lq, H, D = q.shape
    Lk = k.shape[1]
    Lkp = _pad64(Lk)
    if k.stride() != v.stride() or dq.stride() != q.stride() or dk.stride() != k.stride() or dv.stride() != k.stride():
        raise RuntimeError("attn_bwd: stride contract violated")
    with torch.cuda.device(q.device):
        if grad_out.stride(3) != 1 or grad_out.stride(1) % 8 or grad_out.stride(2) % 8:
            grad_out = grad_out.contiguous()
        if not out.is_contiguous():
            raise RuntimeError("attn_bwd: the forward output must be contiguous")
        delta = torch.empty(B, H, Lq, dtype=torch.float32, device=q.device)  # filled by the dQ kernel
        qs, ks, gs = _bhd_strides(q), _bhd_strides(k), _bhd_strides(grad_out)
        _check(_lib.bq_attn_bwd(_p(q), _p(k), _p(v), _p(grad_out), _p(lse), _p(out), _p(delta), _p(mask_log2), _p(dq),
                                _p(dk), _p(dv), B, H, Lq, Lk, Lkp, *qs, *ks, *gs, float(scale), float(p_drop),
                                int(seed) & 0xFFFFFFFF, _p(seed_tensor), int(bool(causal)), _stream()), "attn_bwd")


class _AttnSide(ctypes.Structure):
    _fields_ = [("Q", _vp), ("K", _vp), ("V", _vp), ("dO", _vp), ("O", _vp), ("out", _vp), ("dK", _vp), ("dV", _vp),
                ("LSE", _vp), ("DELTA", _vp), ("mask", _vp), ("Lq", _i), ("Lk", _i), ("Lkp", _i),
                ("q_bs", _l), ("q_rs", _l), ("q_hs", _l), ("k_bs", _l), ("k_rs", _l), ("k_hs", _l),
                ("o_bs", _l), ("o_rs", _l), ("o_hs", _l), ("seed", _u)]


_lib.bq_attn_fwd_pair.argtypes = [ctypes.POINTER(_AttnSide), _i, _i, _f, _f, _vp, _vp]
_lib.bq_attn_fwd_pair.restype = ctypes.c_int
_lib.bq_attn_bwd_pair.argtypes = [ctypes.POINTER(_AttnSide), _i, _i, _f, _f, _vp, _vp]
_lib.bq_attn_bwd_pair.restype = ctypes.c_int


def attn_pair_ok(qa, ka, qb, kb):
    """both attentions qualify for the narrow kernels (bq_attn_fwd_pair): 1..32 queries, more than 128 keys"""
    return qa.shape[1] <= 32 and qb.shape[1] <= 32 and ka.shape[1] > 128 and kb.shape[1] > 128 and qa.shape[0] == qb.shape[0]


def _side(d, q, k, v, lse, mask_log2, seed):
    if k.stride() != v.stride():
        raise RuntimeError("attn pair: k and v must have equal strides")
    d.Q, d.K, d.V, d.LSE, d.mask = _p(q), _p(k), _p(v), _p(lse), _p(mask_log2)
    d.Lq, d.Lk, d.Lkp = q.shape[1], k.shape[1], _pad64(k.shape[1])
    (d.q_bs, d.q_rs, d.q_hs), (d.k_bs, d.k_rs, d.k_hs) = _bhd_strides(q), _bhd_strides(k)
    d.seed = int(seed) & 0xFFFFFFFF


def attn_fwd_pair(sides, scale, p_drop=0.0, seed_tensor=None):
    """two narrow attentions in one launch.  sides: two dicts with q (B, Lq <= 32, H, 64), k / v (B, Lk > 128, H, 64), out
    (B, Lq, H, 64) contiguous bf16 (written), mask_log2 (from key_mask_log2) or None, seed.  Returns the two lse (B, H, Lq)."""
    arr = (_AttnSide * 2)()
    q0 = sides[0]["q"]
    B, H = q0.shape[0], q0.shape[2]
    lses = []
    with torch.cuda.device(q0.device):
        for d, s in zip(arr, sides):
            q, out = s["q"], s["out"]
            if not q.is_cuda or out.shape != q.shape or out.dtype != torch.bfloat16 or not out.is_contiguous():
                raise RuntimeError("attn_fwd_pair: out must be a contiguous bf16 tensor shaped like q (CUDA)")
            lse = torch.empty(B, H, q.shape[1], dtype=torch.float32, device=q.device)
            _side(d, q, s["k"], s["v"], lse, s.get("mask_log2"), s.get("seed", 0))
            d.out = _p(out)
            d.o_bs, d.o_rs, d.o_hs = _bhd_strides(out)
            lses.append(lse)
        _check(_lib.bq_attn_fwd_pair(arr, B, H, float(scale), float(p_drop), _p(seed_tensor), _stream()), "attn_fwd_pair")
    return lses


def attn_bwd_pair(sides, scale, p_drop=0.0, seed_tensor=None):
    """backward of attn_fwd_pair.  sides: q, k, v, out (the forward's, contiguous), lse, grad_out, dq, dk, dv
    (preallocated bf16 views strided like q / k), mask_log2, seed."""
    arr = (_AttnSide * 2)()
    q0 = sides[0]["q"]
    B, H = q0.shape[0], q0.shape[2]
    keep = []
    with torch.cuda.device(q0.device):
        for d, s in zip(arr, sides):
            q, k, go = s["q"], s["k"], s["grad_out"]
            if s["dq"].stride() != q.stride() or s["dk"].stride() != k.stride() or s["dv"].stride() != k.stride():
                raise RuntimeError("attn_bwd_pair: stride contract violated")
            if go.stride(3) != 1 or go.stride(1) % 8 or go.stride(2) % 8:
                go = go.contiguous()
            if not s["out"].is_contiguous():
                raise RuntimeError("attn_bwd_pair: the forward output must be contiguous")
            delta = torch.empty(B, H, q.shape[1], dtype=torch.float32, device=q.device)
            keep += [go, delta]
            _side(d, q, k, s["v"], s["lse"], s.get("mask_log2"), s.get("seed", 0))
            d.dO, d.O, d.DELTA = _p(go), _p(s["out"]), _p(delta)
            d.out, d.dK, d.dV = _p(s["dq"]), _p(s["dk"]), _p(s["dv"])
            d.o_bs, d.o_rs, d.o_hs = _bhd_strides(go)
        _check(_lib.bq_attn_bwd_pair(arr, B, H, float(scale), float(p_drop), _p(seed_tensor), _stream()), "attn_bwd_pair")


_lib.bq_attn_fwd2.argtypes = [_vp] * 8 + [_i] * 6 + [_l] * 12 + [_f, _f, _u, _vp, _vp]
_lib.bq_attn_fwd2.restype = ctypes.c_int
_lib.bq_attn_bwd2.argtypes = [_vp] * 15 + [_i] * 6 + [_l] * 12 + [_f, _f, _u, _vp, _vp]
_lib.bq_attn_bwd2.restype = ctypes.c_int


def key_mask_log2_two(mask, B, Lk1, Lk2):
    """additive key mask broadcastable to (B,1,1,Lk1+Lk2) over cat(segment 1, segment 2) -> f32 (B, pad64(Lk1) +
    pad64(Lk2)) in the two-segment layout of bq_attn_fwd2 (each segment padded to a multiple of 64), times log2(e)"""
    if mask is None:
        return None
    p1 = _pad64(Lk1)
    m = torch.zeros(B, p1 + _pad64(Lk2), dtype=torch.float32, device=mask.device)
    src = mask.reshape(-1, Lk1 + Lk2).expand(B, Lk1 + Lk2).float() * LOG2E
    m[:, :Lk1] = src[:, :Lk1]
    m[:, p1:p1 + Lk2] = src[:, Lk1:]
    return m


def attn_fwd2(q, k1, v1, k2, v2, scale, mask_log2=None, p_drop=0.0, seed=0, seed_tensor=None, out=None):
    """attn_fwd over the keys cat(k1, k2) / values cat(v1, v2) without forming them (Lq <= 32).  k1 / v1 (B, Lk1, H, 64)
    views with equal strides, k2 / v2 (B, Lk2, H, 64) likewise; mask_log2 from key_mask_log2_two."""
    for t, n in ((q, "q"), (k1, "k1"), (v1, "v1"), (k2, "k2"), (v2, "v2")):
        if not t.is_cuda:
            raise RuntimeError("%s: CPU not supported" % n)
    if k1.stride() != v1.stride() or k2.stride() != v2.stride():
        raise RuntimeError("attn_fwd2: k and v of a segment must have equal strides")
    B, Lq, H, D = q.shape
    Lk1, Lk2 = k1.shape[1], k2.shape[1]
    Lkp = _pad64(Lk1) + _pad64(Lk2)
    with torch.cuda.device(q.device):
        if out is None:
            out = torch.empty(B, Lq, H, D, dtype=torch.bfloat16, device=q.device)
        elif out.shape != (B, Lq, H, D) or out.dtype != torch.bfloat16 or not out.is_contiguous():
            raise RuntimeError("attn_fwd2: out must be a contiguous bf16 (B, Lq, H, 64) tensor")
        lse = torch.empty(B, H, Lq, dtype=torch.float32, device=q.device)
        _check(_lib.bq_attn_fwd2(_p(q), _p(k1), _p(v1), _p(k2), _p(v2), _p(out), _p(lse), _p(mask_log2), B, H, Lq, Lk1,
                                 Lk2, Lkp, *_bhd_strides(q), *_bhd_strides(k1), *_bhd_strides(k2), *_bhd_strides(out),
                                 float(scale), float(p_drop), int(seed) & 0xFFFFFFFF, _p(seed_tensor), _stream()),
               "attn_fwd2")
    return out, lse


def attn_bwd2(q, k1, v1, k2, v2, out, lse, grad_out, scale, dq, dk1, dv1, dk2, dv2, mask_log2=None, p_drop=0.0, seed=0,
              seed_tensor=None):
    """Backward of attn_fwd2.  dq strided like q, dk1 / dv1 like k1, dk2 / dv2 like k2 (preallocated bf16 views)."""
    B, Lq, H, D = q.shape
    Lk1, Lk2 = k1.shape[1], k2.shape[1]
    Lkp = _pad64(Lk1) + _pad64(Lk2)
    if (k1.stride() != v1.stride() or k2.stride() != v2.stride() or dq.stride() != q.stride() or
            dk1.stride() != k1.stride() or dv1.stride() != k1.stride() or dk2.stride() != k2.stride() or
            dv2.stride() != k2.stride()):
        raise RuntimeError("attn_bwd2: stride contract violated")
    with torch.cuda.device(q.device):
        if grad_out.stride(3) != 1 or grad_out.stride(1) % 8 or grad_out.stride(2) % 8:
            grad_out = grad_out.contiguous()
        if not out.is_contiguous():
            raise RuntimeError("attn_bwd2: the forward output must be contiguous")
        delta = torch.empty(B, H, Lq, dtype=torch.float32, device=q.device)
        _check(_lib.bq_attn_bwd2(_p(q), _p(k1), _p(v1), _p(k2), _p(v2), _p(grad_out), _p(lse), _p(out), _p(delta),
                                 _p(mask_log2), _p(dq), _p(dk1), _p(dv1), _p(dk2), _p(dv2), B, H, Lq, Lk1, Lk2, Lkp,
                                 *_bhd_strides(q), *_bhd_strides(k1), *_bhd_strides(k2), *_bhd_strides(grad_out),
                                 float(scale), float(p_drop), int(seed) & 0xFFFFFFFF, _p(seed_tensor), _stream()),
               "attn_bwd2")


_lib.bq_colsum_chunks.argtypes = [_i]
_lib.bq_colsum_chunks.restype = ctypes.c_int
_lib.bq_colsum_bf16.argtypes = [_vp, _vp, _i, _i, _vp, _vp, _vp]
_lib.bq_colsum_bf16.restype = ctypes.c_int

_COLSUM_SLOTS = 1 << 16
_colsum_counters = {}
_colsum_next = [0]


def _colsum_counter(device, n):
    """n zeroed, self-resetting completion counters that no other in-flight colsum launch uses (rotating slots)"""
    key = device.index if device.index is not None else torch.cuda.current_device()
    pool = _colsum_counters.get(key)
    if pool is None:
        pool = _colsum_counters[key] = torch.zeros(_COLSUM_SLOTS, dtype=torch.int32, device=device)
    if _colsum_next[0] + n > _COLSUM_SLOTS:
        _colsum_next[0] = 0
    base = _colsum_next[0]
    _colsum_next[0] += n
    return pool.data_ptr() + 4 * base


def colsum(g2d):
    """f32 column sums of a contiguous bf16 (M, N) matrix (bias gradients), N % 4 == 0.  One launch."""
    M, N = g2d.shape
    with torch.cuda.device(g2d.device):
        out = torch.empty(N, dtype=torch.float32, device=g2d.device)
        chunks = _lib.bq_colsum_chunks(M)
        part = cnt = None
        if chunks > 1:
            part = torch.empty(chunks * N, dtype=torch.float32, device=g2d.device)
            cnt = _colsum_counter(g2d.device, (N + 255) // 256)
        _check(_lib.bq_colsum_bf16(_p(g2d), _p(out), M, N, _p(part), cnt, _stream()), "colsum")
    return out


_lib.bq_gelu_fwd_bf16.argtypes = [_vp, _vp, _l, _vp]
_lib.bq_gelu_fwd_bf16.restype = ctypes.c_int


def gelu_fwd(x):
    """exact GELU of a contiguous bf16 tensor (numel % 8 == 0)"""
    with torch.cuda.device(x.device):
        y = torch.empty_like(x)
        _check(_lib.bq_gelu_fwd_bf16(_p(x), _p(y), x.numel(), _stream()), "gelu_fwd")
    return y


# ---- multi-tensor AdamW that also writes the bf16 shadows (csrc/adamw.hip) -----------------------------
_lib.bq_adamw_chunk_elems.restype = ctypes.c_int
_lib.bq_adamw_tensor_bytes.restype = ctypes.c_int
_lib.bq_adamw_multi.argtypes = [_vp, _vp, _i, _vp, _f, _f, _f, _f, _vp]
_lib.bq_adamw_multi.restype = ctypes.c_int
ADAMW_CHUNK = _lib.bq_adamw_chunk_elems()
ADAMW_TENSOR_BYTES = _lib.bq_adamw_tensor_bytes()


def adamw_multi(table, chunks, step, beta1, beta2, eps, grad_clip_value=0.0):
    """table: uint8 device tensor of packed AdamWTensor records; chunks: int32 (n, 2) device tensor; step: f32 device
    scalar holding the 1-based count of THIS update; grad_clip_value > 0: gradients clamped to +-value as they are read."""
    with torch.cuda.device(table.device):
        _check(_lib.bq_adamw_multi(_p(table), _p(chunks), chunks.shape[0], _p(step), float(beta1), float(beta2),
                                   float(eps), float(grad_clip_value or 0.0), _stream()), "adamw")


# ---- multi-tensor bf16 transpose (csrc/transpose.hip): the K-contiguous copies of the text side's weights -------
_lib.bq_transpose_tensor_bytes.restype = ctypes.c_int
_lib.bq_transpose_multi_bf16.argtypes = [_vp, _vp, _i, _i, _vp]
_lib.bq_transpose_multi_bf16.restype = ctypes.c_int
TRANSPOSE_TENSOR_BYTES = _lib.bq_transpose_tensor_bytes()


def transpose_table(pairs, device):
    """pairs: [(src (N, K) bf16 with a contiguous last dim, dst (K, N) bf16 contiguous)], N and K multiples of 64 ->
    (table, chunks) device tensors for transpose_multi"""
    import struct
    assert TRANSPOSE_TENSOR_BYTES == 32
    rec, chunks = [], []
    for i, (src, dst) in enumerate(pairs):
        N, K = src.shape
        if (src.dtype != torch.bfloat16 or dst.dtype != torch.bfloat16 or src.stride(1) != 1 or not dst.is_contiguous()
                or tuple(dst.shape) != (K, N) or N % 64 or K % 64 or src.stride(0) % 8 or src.data_ptr() % 16
                or dst.data_ptr() % 16):
            raise RuntimeError("transpose_table: unsupported pair %s -> %s" % (tuple(src.shape), tuple(dst.shape)))
        rec.append(struct.pack("<QQiiii", src.data_ptr(), dst.data_ptr(), N, K, src.stride(0), K // 64))
        chunks.append(torch.stack([torch.full((N // 64 * (K // 64),), i, dtype=torch.int32),
                                   torch.arange(N // 64 * (K // 64), dtype=torch.int32)], dim=1))
    table = torch.frombuffer(bytearray(b"".join(rec)), dtype=torch.uint8).to(device)
    return table, torch.cat(chunks, dim=0).contiguous().to(device)


def transpose_multi(table, chunks, max_wgs=0):
    """max_wgs > 0: at most that many workgroups walk the tiles (beside a latency-bound chain); 0: one per tile"""
    with torch.cuda.device(table.device):
        _check(_lib.bq_transpose_multi_bf16(_p(table), _p(chunks), chunks.shape[0], int(max_wgs), _stream()), "transpose_multi")


# ---- training-mode BatchNorm + ReLU (+ max over nsample) on point-major bf16 rows (csrc/bn.hip) -------
_lib.bq_bn_chunks.argtypes = [_l, _i, _i]
_lib.bq_bn_chunks.restype = ctypes.c_int
_lib.bq_bn_stats.argtypes = [_vp, _l, _i, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp]
_lib.bq_bn_stats.restype = ctypes.c_int
_lib.bq_bn_apply.argtypes = [_vp, _vp, _vp, _vp, _l, _i, _i, _i, _i, _vp]
_lib.bq_bn_apply.restype = ctypes.c_int
_lib.bq_bn_backward.argtypes = [_vp] * 9 + [_l, _i, _i, _i, _i, _vp]
_lib.bq_bn_backward.restype = ctypes.c_int


def bn_relu_fwd(x, gamma, beta, running_mean, running_var, num_batches_tracked, eps, momentum, S, relu, pool):
    """x: bf16 (R, C) contiguous rows (point-major activations).  Batch statistics (running buffers updated in
    place), then y = relu?(bn(x)) as bf16 (R, C), or (R // S, C) = max over every run of S rows when pool.
    Returns y and the saved statistics (scale, shift, mean, rstd: one f32 (4, C) tensor)."""
    if not x.is_cuda:
        raise RuntimeError("x: CPU not supported")
    R, C = x.shape
    with torch.cuda.device(x.device):
        stats = torch.empty(4, C, dtype=torch.float32, device=x.device)
        part = torch.empty(_lib.bq_bn_chunks(R, 0, 0) * 2 * C, dtype=torch.float32, device=x.device)
        _check(_lib.bq_bn_stats(_p(x), R, C, _p(gamma), _p(beta), _p(running_mean), _p(running_var),
                                _p(num_batches_tracked), float(eps), float(momentum), _p(part), _p(stats[0]),
                                _p(stats[1]), _p(stats[2]), _p(stats[3]), _stream()), "bn_stats")
        y = torch.empty(R // S if pool else R, C, dtype=torch.bfloat16, device=x.device)
        _check(_lib.bq_bn_apply(_p(x), _p(stats[0]), _p(stats[1]), _p(y), R, C, int(S), int(bool(relu)),
                                int(bool(pool)), _stream()), "bn_apply")
    return y, stats


def bn_relu_bwd(dy, x, stats, S, relu, pool):
    """-> dx bf16 (R, C), dgamma f32 (C), dbeta f32 (C)"""
    R, C = x.shape
    with torch.cuda.device(x.device):
        dx = torch.empty_like(x)
        dgb = torch.empty(2, C, dtype=torch.float32, device=x.device)
        part = torch.empty(_lib.bq_bn_chunks(R, int(S), int(bool(pool))) * 2 * C, dtype=torch.float32, device=x.device)
        _check(_lib.bq_bn_backward(_p(dy), _p(x), _p(stats[0]), _p(stats[1]), _p(stats[2]), _p(stats[3]), _p(part),
                                   _p(dgb), _p(dx), R, C, int(S), int(bool(relu)), int(bool(pool)), _stream()),
               "bn_backward")
    return dx, dgb[1], dgb[0]


# ---- fused dropout + residual + LayerNorm (csrc/ln.hip) ---------------------------------------------
def drop_add_ln_fwd(x, residual, gamma, beta, eps, p_drop, seed, seed_tensor, want_sum=False, p_path=0.0,
                    rows_per_sample=0, want_dgb=False):
    """y = LayerNorm(path(dropout(x)) + residual); x, residual bf16 (..., H) contiguous, residual may be None;
    path = stochastic depth per sample of rows_per_sample rows (p_path = 0: identity).
    Returns y, sum (the bf16 pre-norm sum, or None unless want_sum), mean, rstd, dgb (zeroed f32 (2, H) accumulator
    for drop_add_ln_bwd, or None unless want_dgb)."""
    H = x.shape[-1]
    M = x.numel() // H
    with torch.cuda.device(x.device):
        y = torch.empty_like(x)
        s = torch.empty_like(x) if want_sum else None
        mean = torch.empty(M, dtype=torch.float32, device=x.device)
        rstd = torch.empty(M, dtype=torch.float32, device=x.device)
        dgb = torch.empty(2, H, dtype=torch.float32, device=x.device) if want_dgb else None
        _check(_lib.bq_drop_add_ln_fwd(_p(x), _p(residual), _p(gamma), _p(beta), _p(y), _p(s), _p(mean), _p(rstd),
                                       _p(dgb), M, H, float(eps), float(p_drop), float(p_path), int(rows_per_sample),
                                       int(seed) & 0xFFFFFFFF, _p(seed_tensor), _stream()), "drop_add_ln_fwd")
    return y, s, mean, rstd, dgb


def drop_add_ln_bwd(x, residual, gamma, dy, mean, rstd, eps, p_drop, seed, seed_tensor, dsum=None, p_path=0.0,
                    rows_per_sample=0, dgb=None):
    """-> dx, dresidual (None when residual is None), dgamma, dbeta; dsum = gradient of the `sum` output; dgb = the
    zeroed accumulator the forward handed out (a fresh zeros(2, H) when None)"""
    H = x.shape[-1]
    M = x.numel() // H
    with torch.cuda.device(x.device):
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if residual is not None else None
        if dgb is None:
            dgb = torch.zeros(2, H, dtype=torch.float32, device=x.device)
        _check(_lib.bq_drop_add_ln_bwd(_p(x), _p(residual), _p(gamma), _p(dy), _p(dsum), _p(mean), _p(rstd), _p(dx),
                                       _p(dres), _p(dgb), M, H, float(eps), float(p_drop), float(p_path),
                                       int(rows_per_sample), int(seed) & 0xFFFFFFFF, _p(seed_tensor), _stream()),
               "drop_add_ln_bwd")
    return dx, dres, dgb[0], dgb[1]


def drop_add_ln_bwd_sum(s, gamma, dy, mean, rstd, eps, seed, seed_tensor, dsum=None, p_path=0.0, rows_per_sample=0, dgb=None):
    """the backward of a dropout-free site from its stored sum `s` (include/bqhip_fusion.h: bq_drop_add_ln_bwd_sum)
    -> dx, dresidual (the SAME tensor as dx when p_path == 0), dgamma, dbeta"""
    H = s.shape[-1]
    M = s.numel() // H
    with torch.cuda.device(s.device):
        dx = torch.empty_like(s)
        dres = torch.empty_like(s) if p_path > 0 else None
        if dgb is None:
            dgb = torch.zeros(2, H, dtype=torch.float32, device=s.device)
        _check(_lib.bq_drop_add_ln_bwd_sum(_p(s), _p(gamma), _p(dy), _p(dsum), _p(mean), _p(rstd), _p(dx), _p(dres), _p(dgb),
                                           M, H, float(eps), float(p_path), int(rows_per_sample), int(seed) & 0xFFFFFFFF,
                                           _p(seed_tensor), _stream()), "drop_add_ln_bwd_sum")
    return dx, (dres if dres is not None else dx), dgb[0], dgb[1]


def twin_drop_add_ln_fwd(x, residual, gamma, beta, gamma2, beta2, eps, p_drop, seed, seed_tensor, want_dgb=False):
    """drop_add_ln_fwd over two row groups (first / second half of the rows of x) with their own LayerNorm parameters;
    returns y, mean, rstd, dgb (zeroed f32 (2, 2, H), or None)"""
    H = x.shape[-1]
    M = x.numel() // H
    with torch.cuda.device(x.device):
        y = torch.empty_like(x)
        mean = torch.empty(M, dtype=torch.float32, device=x.device)
        rstd = torch.empty(M, dtype=torch.float32, device=x.device)
        dgb = torch.empty(2, 2, H, dtype=torch.float32, device=x.device) if want_dgb else None
        _check(_lib.bq_twin_drop_add_ln_fwd(_p(x), _p(residual), _p(gamma), _p(beta), _p(gamma2), _p(beta2), _p(y),
                                            _p(mean), _p(rstd), _p(dgb), M, H, float(eps), float(p_drop),
                                            int(seed) & 0xFFFFFFFF, _p(seed_tensor), _stream()), "twin_drop_add_ln_fwd")
    return y, mean, rstd, dgb


def twin_drop_add_ln_bwd(x, residual, gamma, gamma2, dy, mean, rstd, eps, p_drop, seed, seed_tensor, dgb=None):
    """-> dx, dresidual, dgb (2, 2, H): [group][dgamma | dbeta]"""
    H = x.shape[-1]
    M = x.numel() // H
    with torch.cuda.device(x.device):
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if residual is not None else None
        if dgb is None:
            dgb = torch.zeros(2, 2, H, dtype=torch.float32, device=x.device)
        _check(_lib.bq_twin_drop_add_ln_bwd(_p(x), _p(residual), _p(gamma), _p(gamma2), _p(dy), _p(mean), _p(rstd),
                                            _p(dx), _p(dres), _p(dgb), M, H, float(eps), float(p_drop),
                                            int(seed) & 0xFFFFFFFF, _p(seed_tensor), _stream()), "twin_drop_add_ln_bwd")
    return dx, dres, dgb


# ---- proposal post-processing (csrc/nms.hip) ------------------------------------------------------------
_lib.bq_box_point_count.argtypes = [_vp] * 5 + [_i] * 5 + [_vp]
_lib.bq_box_point_count.restype = ctypes.c_int
_lib.bq_nms.argtypes = [_vp] * 5 + [_i, _i, _f, _i, _i, _vp]
_lib.bq_nms.restype = ctypes.c_int


def box_point_count(points, center, size, heading, cap=0):
    """points f32 (B, N, >=3) (xyz first), center / size f32 (B, K, 3), heading f32 (B, K) -> i32 (B, K) points inside each
    oriented box (closed)"""
    for t, n in ((points, "points"), (center, "center"), (size, "size"), (heading, "heading")):
        if not t.is_cuda:
            raise RuntimeError("%s: CPU not supported" % n)
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise RuntimeError("%s must be a contiguous float32 tensor" % n)
    B, N, ld = points.shape
    K = center.shape[1]
    with torch.cuda.device(points.device):
        out = torch.empty(B, K, dtype=torch.int32, device=points.device)
        _check(_lib.bq_box_point_count(_p(points), _p(center), _p(size), _p(heading), _p(out), B, N, ld, K, int(cap),
                                       _stream()), "box_point_count")
    return out


def nms(box, score, thresh, cls=None, valid=None, old_type=False, same_cls=False):
    """box f32 (B, K, 6) axis-aligned extents, score f32 (B, K), cls i32 (B, K), valid u8 / bool (B, K) -> bool (B, K)"""
    if not box.is_cuda:
        raise RuntimeError("box: CPU not supported")
    B, K = score.shape
    box, score = box.contiguous().float(), score.contiguous().float()
    cls = cls.contiguous().to(torch.int32) if cls is not None else None
    valid = valid.contiguous().to(torch.uint8) if valid is not None else None
    with torch.cuda.device(box.device):
        keep = torch.empty(B, K, dtype=torch.uint8, device=box.device)
        _check(_lib.bq_nms(_p(box), _p(score), _p(cls), _p(valid), _p(keep), B, K, float(thresh), int(bool(old_type)),
                           int(bool(same_cls)), _stream()), "nms")
    return keep.bool()


# ---- multiview projection (csrc/projection.hip) ---------------------------------------------------------
_lib.bq_project_points.argtypes = [_vp] * 4 + [_i] * 4 + [_f] * 7 + [_vp]
_lib.bq_project_points.restype = ctypes.c_int
_lib.bq_fuse_point_features.argtypes = [_vp] * 3 + [_i] * 5 + [_vp]
_lib.bq_fuse_point_features.restype = ctypes.c_int


def project_points(points, depth, frames, image_dims, fx, fy, cx, cy, depth_min, depth_max, accuracy):
    """points f32 (N, 3), depth f32 (F, H, W), frames f32 (F, 40) -> i32 (F, N): pixel index y * W + x or -1"""
    for t, n in ((points, "points"), (depth, "depth"), (frames, "frames")):
        if not t.is_cuda:
            raise RuntimeError("%s: CPU not supported" % n)
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise RuntimeError("%s must be a contiguous float32 tensor" % n)
    W, H = int(image_dims[0]), int(image_dims[1])
    F, N = depth.shape[0], points.shape[0]
    if depth.numel() != F * W * H or frames.shape != (F, 40) or points.shape[1] != 3:
        raise RuntimeError("project_points: shapes %s %s %s" % (tuple(points.shape), tuple(depth.shape), tuple(frames.shape)))
    with torch.cuda.device(points.device):
        pix = torch.empty(F, N, dtype=torch.int32, device=points.device)
        _check(_lib.bq_project_points(_p(points), _p(depth), _p(frames), _p(pix), F, N, W, H, float(fx), float(fy), float(cx),
                                      float(cy), float(depth_min), float(depth_max), float(accuracy), _stream()),
               "project_points")
    return pix


def fuse_point_features(pix, feat, maxpool):
    """pix i32 (F, N), feat f32 (F, H*W, C) pixel-major -> f32 (N, C)"""
    if not (pix.is_cuda and feat.is_cuda):
        raise RuntimeError("fuse_point_features: CPU not supported")
    if pix.dtype != torch.int32 or feat.dtype != torch.float32 or not pix.is_contiguous() or not feat.is_contiguous():
        raise RuntimeError("fuse_point_features: pix must be contiguous int32, feat contiguous float32")
    F, N = pix.shape
    if feat.dim() != 3 or feat.shape[0] != F:
        raise RuntimeError("fuse_point_features: feat must be (F, H*W, C)")
    HW, C = feat.shape[1], feat.shape[2]
    with torch.cuda.device(pix.device):
        out = torch.empty(N, C, dtype=torch.float32, device=pix.device)
        _check(_lib.bq_fuse_point_features(_p(pix), _p(feat), _p(out), F, N, HW, C, int(bool(maxpool)), _stream()),
               "fuse_point_features")
    return out


# ---- MFMA bf16 GEMM family (csrc/gemm.hip) ------------------------------------------------------------
GEMM_P_XC, GEMM_Q_XC, GEMM_OUT_F32, GEMM_BACKGROUND = 1, 2, 4, 8
EPI_NONE, EPI_BIAS, EPI_BIAS_GELU, EPI_DGELU, EPI_BIAS_CE, EPI_ADD = 0, 1, 2, 3, 4, 5


class _GemmDesc(ctypes.Structure):
    _fields_ = [("P", _vp), ("Q", _vp), ("out", _vp), ("bias", _vp), ("out2", _vp), ("aux", _vp), ("colsum", _vp),
                ("ldp", _i), ("ldq", _i), ("ldo", _i), ("Ni", _i), ("Nj", _i), ("Kc", _i), ("bias_bf16", _i), ("p_bytes", _l), ("q_bytes", _l), ("ksplit", _i),
                ("q_rpb", _i), ("q_bstride", _i), ("o_rpb", _i), ("o_bstride", _i), ("accum", _i)]


_lib.bq_gemm_bf16.argtypes = [ctypes.POINTER(_GemmDesc), _i, _i, _i, _i, _vp]
_lib.bq_gemm_bf16.restype = ctypes.c_int
_lib.bq_gemm_max_problems.restype = ctypes.c_int
GEMM_TILE_ROWS = 1024  # problems with at least this many j rows (and i columns >= 256) run on the large-tile kernels
# Large problems whose output is bf16 with a K-contiguous Q (forward, input gradient) take the 256 x 128 persistent kernel
# (csrc/gemm_mid.hip: two workgroups per CU; round 3 A/B against the 256 x 256 one: 38.8 vs 39.2 ms per c3 step).


def _mat(t, name):
    if not t.is_cuda:
        raise RuntimeError("%s: CPU not supported (bridgeqa_amd has no CPU path)" % name)
    if t.dim() != 2 or t.stride(1) != 1:
        raise RuntimeError("%s must be a 2-D tensor with a contiguous last dimension" % name)
    return t


def _mat_rows(t, name):
    """Q / out / aux / out2 of a GEMM problem: a 2-D (rows, cols) tensor, or a 3-D (batch, rows_per_batch, cols) VIEW whose
    batches are stride(0) elements apart -- the batched-row map of bq_gemm_desc (q_rpb / o_rpb), e.g. kv[:, :1025] of a
    (B, 1045, 1536) key/value tensor.  Returns (shape2d, ld, rpb, bstride); a 3-D tensor that is plain rows comes back
    with rpb = 0."""
    if not t.is_cuda:
        raise RuntimeError("%s: CPU not supported (bridgeqa_amd has no CPU path)" % name)
    if t.dim() == 2:
        if t.stride(1) != 1:
            raise RuntimeError("%s must have a contiguous last dimension" % name)
        return (t.shape[0], t.shape[1]), t.stride(0), 0, 0
    if t.dim() != 3 or t.stride(2) != 1:
        raise RuntimeError("%s must be (rows, cols) or a (batch, rows, cols) view with a contiguous last dimension" % name)
    B, R, C = t.shape
    ld, bs = t.stride(1), t.stride(0)
    if B == 1 or bs == R * ld:
        return (B * R, C), ld, 0, 0
    if bs < R * ld or bs % 8 or R > 65535:
        raise RuntimeError("%s: unsupported batched-row view (stride %s, shape %s)" % (name, t.stride(), tuple(t.shape)))
    return (B * R, C), ld, R, bs


# forward / dX launches (K-contiguous operands, plain or bias epilogue) whose contraction is at least this long run on the
# 256 x 256 kernel instead of the 256 x 128 one: since round 6's K loop (two 32-MFMA phases, no compiler fence) it is level or
# ahead alone on fc2 forward (71 against 77 us), dX through fc1 (71 / 72) and dX through qkv (56 / 56), and inside the c3 step
# 31.81 against 32.05 ms (five interleaved pairs, 7 of 8 pairs over two calls in favour; profiles/r06_tile256_long_k.txt).
# 0 = never (tools/ab_bench.py "_ext.TILE256_MIN_K[0]=0")
TILE256_MIN_K = [2304]


def pick_tile(Ni, Nj, q_xc, mid_ok=False):
    if Nj >= GEMM_TILE_ROWS and Ni >= 256:
        return 128 if mid_ok else 256
    if q_xc or Nj > 512:  # (64-row tiles from 100 / 200 rows up: measured 0.3-0.4 ms SLOWER per c3 step)
        return 64
    return 32


def gemm_grouped(problems, flags, epilogue=EPI_NONE, tile=None):
    """One launch for a list of problems out[j][i] = epilogue(sum_kc P(i,kc) Q(j,kc)) (include/bqhip_fusion.h,
    bq_gemm_bf16).  problems: dicts with P, Q, out (2-D tensors, contiguous last dim) and optional bias (fp32 (Ni,)),
    out2, aux, colsum (fp32 (Ni,), accumulated).  P: (Ni, Kc), or (Kc, Ni) with GEMM_P_XC; Q likewise over j."""
    n = len(problems)
    arr = (_GemmDesc * n)()
    pxc, qxc, f32 = bool(flags & GEMM_P_XC), bool(flags & GEMM_Q_XC), bool(flags & GEMM_OUT_F32)
    t_auto = 32
    any_map = False
    for k, pr in enumerate(problems):
        P, Q, out = _mat(pr["P"], "P"), pr["Q"], pr["out"]
        qshape, ldq, q_rpb, q_bs = _mat_rows(Q, "Q")
        oshape, ldo, o_rpb, o_bs = _mat_rows(out, "out")
        any_map = any_map or bool(q_rpb or o_rpb)
        if P.dtype != torch.bfloat16 or Q.dtype != torch.bfloat16:
            raise RuntimeError("gemm: operands must be bf16")
        if out.dtype != (torch.float32 if f32 else torch.bfloat16):
            raise RuntimeError("gemm: out must be %s" % ("float32" if f32 else "bfloat16"))
        Ni, Kc = (P.shape[1], P.shape[0]) if pxc else (P.shape[0], P.shape[1])
        Nj, Kq = (qshape[1], qshape[0]) if qxc else (qshape[0], qshape[1])
        Ni = int(pr.get("Ni", Ni))  # output wider than P's rows (padded vocabulary): P's true extent goes in p_bytes
        if (Kq != Kc and "Kc" not in pr) or tuple(oshape) != (Nj, Ni):
            raise RuntimeError("gemm: shape mismatch P%s Q%s out%s" % (tuple(P.shape), tuple(Q.shape), tuple(out.shape)))
        d = arr[k]
        d.P, d.Q, d.out = P.data_ptr(), Q.data_ptr(), out.data_ptr()
        bias, out2, aux, colsum = pr.get("bias"), pr.get("out2"), pr.get("aux"), pr.get("colsum")
        if bias is not None and (bias.dtype not in (torch.float32, torch.bfloat16) or bias.numel() != Ni
                                 or not bias.is_contiguous()):
            raise RuntimeError("gemm: bias must be a contiguous fp32 or bf16 (Ni,) tensor")
        d.bias_bf16 = int(bias is not None and bias.dtype == torch.bfloat16)
        # (weight-gradient form: colsum (Nj,) receives the column sums of Q over the contraction; with ksplit > 1 it is
        # accumulated with atomics and must arrive zeroed)
        n_cs = Nj if (pxc and qxc and f32) else Ni
        if epilogue != EPI_BIAS_CE and colsum is not None and (colsum.dtype != torch.float32 or colsum.numel() != n_cs
                                                               or not colsum.is_contiguous()):
            raise RuntimeError("gemm: colsum must be a contiguous fp32 (%d,) tensor" % n_cs)
        for t, nm in ((out2, "out2"), (aux, "aux")):
            if epilogue != EPI_BIAS_CE and t is not None and (t.dtype != torch.bfloat16 or t.shape != out.shape
                                                              or t.stride() != out.stride()):
                raise RuntimeError("gemm: %s must be bf16 and laid out like out" % nm)
        d.bias, d.out2, d.aux, d.colsum = _p(bias), _p(out2), _p(aux), _p(colsum)
        d.ldp, d.ldq, d.ldo = P.stride(0), ldq, ldo
        d.q_rpb, d.q_bstride, d.o_rpb, d.o_bstride = q_rpb, q_bs, o_rpb, o_bs
        d.accum = int(bool(pr.get("accum", False)))
        d.Ni, d.Nj, d.Kc = Ni, Nj, Kc
        d.p_bytes, d.q_bytes, d.ksplit = int(pr.get("p_bytes", 0)), int(pr.get("q_bytes", 0)), int(pr.get("ksplit", 1))
        if "Kc" in pr:  # contraction longer than the K-contiguous operand's rows (its partner is zero-padded)
            d.Kc = int(pr["Kc"])
        # (the 256 x 128 kernel also has a weight-gradient form -- tile=128 on request: measured equal to the 256 x 256 kernel
        # alone and slower inside the step, so it is never the automatic choice)
        mid_ok = (not qxc and not f32 and d.Kc >= 128 and colsum is None
                  and epilogue in (EPI_NONE, EPI_BIAS, EPI_BIAS_GELU, EPI_DGELU, EPI_ADD))
        all_mid_ok = mid_ok if k == 0 else (all_mid_ok and mid_ok)
        t_auto = max(t_auto, pick_tile(Ni, Nj, qxc, True))
    if t_auto == 128 and not all_mid_ok:
        t_auto = 256   # (one tile class per launch: every problem of the group must be able to take the 256 x 128 kernel)
    if (t_auto == 128 and TILE256_MIN_K[0] and not pxc and epilogue in (EPI_NONE, EPI_BIAS)
            and all(int(arr[k].Kc) >= TILE256_MIN_K[0] for k in range(n))):
        t_auto = 256   # (long contractions: TILE256_MIN_K above)
    if t_auto == 256 and any_map and tile is None and not (pxc and qxc and f32):
        t_auto = 64    # (the 256 x 256 kernel maps only the contraction rows of its weight-gradient form)
    dev = problems[0]["out"].device
    if (tile or t_auto) != 128:
        flags &= ~GEMM_BACKGROUND   # (only the persistent 256 x 128 kernel has a reduced grid)
    with torch.cuda.device(dev):
        if (tile or t_auto) == 128 and n == 1 and (STREAMK[0] or STREAMK256[0]):
            _streamk_workspace(dev)
        _check(_lib.bq_gemm_bf16(arr, n, int(flags), int(epilogue), int(tile or t_auto), _stream()), "gemm_bf16")


# ---- stream-K workspaces (bq_gemm_set_workspace): one per (device, stream), allocated at the first tile-128 single-problem
# launch on that stream OUTSIDE a capture (the warm-up steps of pipeline.py / graphed.py run on the phase streams first); a
# launch on a stream without one runs on whole tiles
STREAMK = [False]   # the 256 x 128 kernel's stream-K form: built, parity-green, SLOWER on every shape measured
# the 256 x 256 kernel's stream-K form (the vendor library's design for these shapes): built at the end of round 5, parity-green,
# SLOWER too (fc2 forward 98.6 us against 76.7 on whole 256 x 256 tiles and 75.4 on 256 x 128; c3 + 0.9 ms) -- off;
# BQ_GEMM_STREAMK256=1 switches it on
STREAMK256 = [os.environ.get("BQ_GEMM_STREAMK256", "0") == "1"]
_SK_WS = {}
_lib.bq_gemm_workspace_bytes.restype = ctypes.c_long
_lib.bq_gemm_set_workspace.argtypes = [_vp, ctypes.c_long, _vp]
_lib.bq_gemm_set_workspace.restype = ctypes.c_int
_lib.bq_gemm_streamk_mode.argtypes = [_i]
_lib.bq_gemm_streamk_mode.restype = ctypes.c_int


def _streamk_apply():
    on = STREAMK[0] or STREAMK256[0]
    _lib.bq_gemm_streamk_mode((1 if STREAMK[0] else 0) | (2 if STREAMK256[0] else 0))
    for (d, sp), (ws, st) in _SK_WS.items():
        with torch.cuda.device(d):
            _check(_lib.bq_gemm_set_workspace(_p(ws) if on else None, ws.numel() if on else 0, sp), "gemm_set_workspace")


def streamk_enable(flag):
    """measurement / test switch of the 256 x 128 kernel's stream-K form (with both forms off every registered workspace is
    withdrawn and the launches run on whole tiles)"""
    STREAMK[0] = bool(flag)
    _streamk_apply()


def streamk256_enable(flag):
    """the same for the 256 x 256 kernel's form (product default: off as well)"""
    STREAMK256[0] = bool(flag)
    _streamk_apply()


_streamk_apply()


def _streamk_workspace(dev):
    st = torch.cuda.current_stream(dev)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), st.cuda_stream)
    if key in _SK_WS or torch.cuda.is_current_stream_capturing():
        return
    nbytes = int(_lib.bq_gemm_workspace_bytes())
    ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)     # (the tickets at its head must start at zero)
    _check(_lib.bq_gemm_set_workspace(_p(ws), nbytes, st.cuda_stream), "gemm_set_workspace")
    _SK_WS[key] = (ws, st)      # (the stream object is kept: its handle must not be recycled for another stream)


def gemm_fwd(x, w, bias=None, gelu=False, tile=None, out=None, background=False):
    """y = x @ w^T + bias (x (M,K), w (N,K) bf16; bias fp32 (N,)); gelu: returns (y_pre, gelu(y_pre)); out: write y there;
    background: GEMM_BACKGROUND (a side-stream launch that leaves room on every CU)"""
    _mat(x, "x"), _mat(w, "w")
    M, N = x.shape[0], w.shape[0]
    with torch.cuda.device(x.device):
        y = torch.empty(M, N, dtype=torch.bfloat16, device=x.device) if out is None else out
        act = torch.empty_like(y) if gelu else None
    epi = EPI_BIAS_GELU if gelu else (EPI_BIAS if bias is not None else EPI_NONE)
    gemm_grouped([dict(P=w, Q=x, out=y, bias=bias, out2=act)], GEMM_BACKGROUND if background else 0, epi, tile)
    return (y, act) if gelu else y


def gemm_dx(dy, w, pre_act=None, colsum=None, tile=None, add=None, background=False, wt=None):
    """dx = dy @ w (dy (M,N), w (N,K) bf16) [* gelu'(pre_act) (M,K)] [+ add (M,K) bf16]; colsum (K,) fp32 += column
    sums of dx.  wt: w's transpose (K,N), contiguous -- the launch then reads it K-contiguous like a forward (the small-M
    launches of the text side: fusion_state.transposed_shadow) instead of walking w's rows contraction-major"""
    _mat(dy, "dy"), _mat(w, "w")
    M, K = dy.shape[0], w.shape[1]
    if pre_act is not None and add is not None:
        raise RuntimeError("gemm_dx: pre_act and add are exclusive")
    if wt is not None and (colsum is not None or tuple(wt.shape) != (K, w.shape[0])):
        wt = None
    with torch.cuda.device(dy.device):
        dx = torch.empty(M, K, dtype=torch.bfloat16, device=dy.device)
    gemm_grouped([dict(P=w if wt is None else wt, Q=dy, out=dx, aux=pre_act if add is None else add, colsum=colsum)],
                 (GEMM_P_XC if wt is None else 0) | (GEMM_BACKGROUND if background else 0),
                 EPI_DGELU if pre_act is not None else (EPI_ADD if add is not None else EPI_NONE), tile)
    return dx


def gemm_dw(dy, x, tile=None):
    """dw = dy^T @ x in fp32 (dy (M,N), x (M,K) bf16) -> (N,K)"""
    _mat(dy, "dy"), _mat(x, "x")
    N, K = dy.shape[1], x.shape[1]
    with torch.cuda.device(dy.device):
        dw = torch.empty(N, K, dtype=torch.float32, device=dy.device)
    gemm_grouped([dict(P=x, Q=dy, out=dw)], GEMM_P_XC | GEMM_Q_XC | GEMM_OUT_F32, EPI_NONE, tile)
    return dw


class _ColsumDesc(ctypes.Structure):
    _fields_ = [("g", _vp), ("out", _vp), ("M", _i), ("N", _i), ("ld", _i)]


_lib.bq_colsum_grouped_bf16.argtypes = [ctypes.POINTER(_ColsumDesc), _i, _vp]
_lib.bq_colsum_grouped_bf16.restype = ctypes.c_int


def colsum_grouped(mats):
    """fp32 column sums of several bf16 (M_p, N_p) matrices (contiguous last dim) in one launch; returns a list of
    (N_p,) views of ONE zero-initialised buffer (one memset + one kernel for all bias gradients of a backward pass)"""
    n = len(mats)
    dev = mats[0].device
    with torch.cuda.device(dev):
        flat = torch.zeros(sum(m.shape[1] for m in mats), dtype=torch.float32, device=dev)
        arr = (_ColsumDesc * n)()
        outs, off = [], 0
        for k, m in enumerate(mats):
            _mat(m, "g")
            if m.dtype != torch.bfloat16:
                raise RuntimeError("colsum_grouped: bf16 only")
            o = flat[off:off + m.shape[1]]
            off += m.shape[1]
            outs.append(o)
            arr[k].g, arr[k].out, arr[k].M, arr[k].N, arr[k].ld = m.data_ptr(), o.data_ptr(), m.shape[0], m.shape[1], m.stride(0)
        _check(_lib.bq_colsum_grouped_bf16(arr, n, _stream()), "colsum_grouped")
    return outs


_lib.bq_wgrad_rows_supported.argtypes = [_i, _i]
_lib.bq_wgrad_rows_supported.restype = ctypes.c_int
_lib.bq_wgrad_rows_workgroups.argtypes = [_l, _i, _i, _i]
_lib.bq_wgrad_rows_workgroups.restype = ctypes.c_int
_lib.bq_wgrad_rows_bf16.argtypes = [_vp, _vp, _vp, _vp, _l, _i, _i, _i, _i, _i, _i, _vp]
_lib.bq_wgrad_rows_bf16.restype = ctypes.c_int


def wgrad_rows_ok(Ni, Nj):
    return bool(_lib.bq_wgrad_rows_supported(int(Ni), int(Nj)))


def wgrad_rows(x, dy, out, workgroups=0):
    """out (Nj, ldo) fp32 = dy^T x over whole rows: x (R, Ni) bf16 rows of stride ldp, dy (R, Nj) bf16; every operand row
    read once, per-workgroup partial products summed by a second kernel (bq_wgrad_rows_bf16; the detector's SharedMLP
    weight gradients).  out must be a whole (Nj, ldo) buffer: its columns [Ni, ldo) are zeroed."""
    _mat(x, "x"), _mat(dy, "dy")
    if out.dtype != torch.float32 or out.stride(1) != 1 or x.shape[0] != dy.shape[0] or out.shape[0] != dy.shape[1]:
        raise RuntimeError("wgrad_rows: out must be fp32 (Nj, ldo) rows, x and dy the same number of rows")
    with torch.cuda.device(x.device):
        w = _lib.bq_wgrad_rows_workgroups(x.shape[0], x.shape[1], dy.shape[1], int(workgroups))
        part = torch.empty(w * dy.shape[1] * out.stride(0), dtype=torch.float32, device=x.device)
        _check(_lib.bq_wgrad_rows_bf16(_p(x), _p(dy), _p(out), _p(part), x.shape[0], x.shape[1], dy.shape[1], x.stride(0),
                                       dy.stride(0), out.stride(0), int(workgroups), _stream()), "wgrad_rows")
    return out


# ---- SharedMLP layer on point-major rows: 1x1 convolution + BatchNorm statistics in its epilogue (csrc/gemm.hip) ----
_lib.bq_pwconv_records.argtypes = [_l, _i]
_lib.bq_pwconv_records.restype = ctypes.c_int
_lib.bq_pwconv_bn_fwd.argtypes = [_vp, _l, _i, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp,
                                  _vp, _vp, _vp]
_lib.bq_pwconv_bn_fwd.restype = ctypes.c_int


def pwconv_bn_relu_fwd(x, K, w_pad, gamma, beta, running_mean, running_var, num_batches_tracked, eps, momentum, S, relu,
                       pool, center=None, want_arg=False, x_stats=None, defer_apply=False):
    """x: bf16 rows (R, ldx) view with contiguous elements (ldx = x.stride(0) >= K, the first K of a row are the input
    channels); w_pad: bf16 (N, Kc) contiguous, zero beyond K, Kc % 64 == 0.  y_raw = x @ w^T (bf16 (R, N)), its
    training-mode BatchNorm statistics from the fp32 accumulators (running buffers updated in place), then
    out = relu?(bn(y_raw)) as bf16 (R, N), or (R // S, N) = max over every run of S rows when pool.
    center (f32 (N,), e.g. running_mean itself): y_raw holds x @ w^T - center and stats describe the stored values (same
    `out`; the bf16 rounding of y_raw then applies to the deviation from the channel mean -- bq_pwconv_bn_fwd).
    Returns out, y_raw, stats (f32 (4, N): scale, shift, mean, rstd); with want_arg (pooled layers) a fourth value: u8
    (R // S, N), the row of each group's first maximum (None where the fp32-accumulator apply pass produced the output).
    Deferred activations between the layers of a SharedMLP (bq_pwconv_bn_fwd_x): x_stats = the previous layer's stats when x
    is that layer's y_raw (the input is then relu(x * x_stats[0] + x_stats[1]), applied tile by tile inside the kernel);
    defer_apply: skip this layer's own BatchNorm + ReLU pass -- out is None and the next layer reads y_raw."""
    if not x.is_cuda:
        raise RuntimeError("x: CPU not supported")
    R, N, Kc = x.shape[0], w_pad.shape[0], w_pad.shape[1]
    arg = None
    with torch.cuda.device(x.device):
        y_raw = torch.empty(R, N, dtype=torch.bfloat16, device=x.device)
        stats = torch.empty(5, N, dtype=torch.float32, device=x.device)   # scale, shift, mean, rstd (of the stored y) | shift_acc
        part = torch.empty(_lib.bq_pwconv_records(R, N) * 3 * N, dtype=torch.float32, device=x.device)
        _check(_lib.bq_pwconv_bn_fwd_x(_p(x), _p(x_stats[0]) if x_stats is not None else None,
                                       _p(x_stats[1]) if x_stats is not None else None, R, int(K), x.stride(0), _p(w_pad),
                                       w_pad.stride(0), Kc, N, _p(y_raw), _p(part), _p(gamma), _p(beta), _p(running_mean),
                                       _p(running_var), _p(num_batches_tracked), float(eps), float(momentum), _p(stats[0]),
                                       _p(stats[1]), _p(stats[2]), _p(stats[3]), _p(center), _p(stats[4]), _stream()),
               "pwconv_bn_fwd")
        if defer_apply:
            return (None, y_raw, stats, None) if want_arg else (None, y_raw, stats)
        out = torch.empty(R // S if pool else R, N, dtype=torch.bfloat16, device=x.device)
        if FP32_PREACT[0] and (not pool or (int(S) in (16, 32, 64) and R % int(S) == 0)):
            # BatchNorm + ReLU (+ max-pool) on the fp32 accumulators of the product computed once more: the output never
            # passes through the bf16 y_raw (which the backward still reads)
            _check(_lib.bq_pwconv_bn_apply(_p(x), R, int(K), x.stride(0), _p(w_pad), w_pad.stride(0), Kc, N, _p(stats[0]),
                                           _p(stats[4]), _p(out), int(S), int(bool(relu)), int(bool(pool)),
                                           _stream()), "pwconv_bn_apply")
        else:
            if want_arg and pool and int(S) <= 256:
                arg = torch.empty(R // S, N, dtype=torch.uint8, device=x.device)
            _check(_lib.bq_bn_apply_arg(_p(y_raw), _p(stats[0]), _p(stats[1]), _p(out), _p(arg), R, N, int(S),
                                        int(bool(relu)), int(bool(pool)), _stream()), "bn_apply")
    return (out, y_raw, stats, arg) if want_arg else (out, y_raw, stats)


# ---- the backward of a SharedMLP layer in one pass over its activations (csrc/detbwd.hip) ------------------------------------
_lib.bq_bn_apply_arg.argtypes = [_vp, _vp, _vp, _vp, _vp, _l, _i, _i, _i, _i, _vp]
_lib.bq_bn_apply_arg.restype = ctypes.c_int
_lib.bq_bn_backward_reduce.argtypes = [_vp] * 8 + [_l, _i, _i, _i, _i, _vp]
_lib.bq_bn_backward_reduce.restype = ctypes.c_int
_lib.bq_bn_backward_reduce_arg.argtypes = [_vp] * 9 + [_l, _i, _i, _i, _vp]
_lib.bq_bn_backward_reduce_arg.restype = ctypes.c_int
_lib.bq_sa_bwd_supported.argtypes = [_i, _i, _i, _i, _i]
_lib.bq_sa_bwd_supported.restype = ctypes.c_int
_lib.bq_sa_bwd_workgroups.argtypes = [_l, _i, _i, _i, _i]
_lib.bq_sa_bwd_workgroups.restype = ctypes.c_int
_lib.bq_sa_bwd_fused.argtypes = [_vp] * 13 + [_l, _i, _i, _i, _i, _i, _i, _i, _vp]
_lib.bq_sa_bwd_fused.restype = ctypes.c_int
_lib.bq_sa_bwd_fused_x.argtypes = [_vp] * 15 + [_l, _i, _i, _i, _i, _i, _i, _i, _vp]
_lib.bq_sa_bwd_fused_x.restype = ctypes.c_int
_lib.bq_sa_bwd_fused_xr.argtypes = [_vp] * 19 + [_l, _i, _i, _i, _i, _i, _i, _i, _vp]
_lib.bq_sa_bwd_fused_xr.restype = ctypes.c_int
_lib.bq_sa_bwd_reduce_supported.argtypes = [_i, _i, _i, _i]
_lib.bq_sa_bwd_reduce_supported.restype = ctypes.c_int
# a layer's fused backward can also sum the PREVIOUS layer's dbeta / dgamma (its dOut and its pre-activation are both in the pass:
# "the BatchNorm reduce in the epilogue of the next layer's dX GEMM").  Built, parity-green, and SLOWER: the second barrier per
# tile, the table reads and 32 more live registers per lane cost the fused kernels more than the eight reduction launches they
# replace (c2 7.72 / 7.77 -> 7.91 / 7.95 ms, c3 33.50 -> 33.41 inside the noise).  Off; BQ_CARRY_REDUCE=1 switches it on.
CARRY_REDUCE = [os.environ.get("BQ_CARRY_REDUCE", "0") == "1"]
_lib.bq_pwconv_bn_fwd_x.argtypes = [_vp, _vp, _vp, _l, _i, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp,
                                    _vp, _vp, _vp, _vp, _vp]
_lib.bq_pwconv_bn_fwd_x.restype = ctypes.c_int
# SharedMLP layers hand their stored pre-activation to the next layer, which applies BatchNorm + ReLU on load (no bn_apply pass,
# no activation tensor between two convolutions); off / BQ_DEFER_BN=0: every layer writes its activation
DEFER_BN = [os.environ.get("BQ_DEFER_BN", "1") != "0"]
# SharedMLP backward as reduction + one fused pass (off, or BQ_FUSED_SA_BWD=0 for A/B runs: bn_relu_bwd + dX GEMM + wgrad_rows)
FUSED_SA_BWD = [os.environ.get("BQ_FUSED_SA_BWD", "1") != "0"]


def bn_apply(y_raw, stats, S, relu, pool):
    """relu?(y_raw * stats[0] + stats[1]) as bf16 (R, N), or the max over runs of S rows (R // S, N) when pool"""
    R, N = y_raw.shape
    with torch.cuda.device(y_raw.device):
        out = torch.empty(R // S if pool else R, N, dtype=torch.bfloat16, device=y_raw.device)
        _check(_lib.bq_bn_apply(_p(y_raw), _p(stats[0]), _p(stats[1]), _p(out), R, N, int(S), int(bool(relu)),
                                int(bool(pool)), _stream()), "bn_apply")
    return out


def sa_bwd_supported(ldx, N, S, pool, need_dx):
    return bool(_lib.bq_sa_bwd_supported(int(ldx), int(N), int(S), int(bool(pool)), int(bool(need_dx))))


def bn_bwd_reduce(dy, x, stats, S, relu, pool, arg=None):
    """dgb f32 (2, C) = dbeta | dgamma of bn_relu_bwd alone; arg (pooled layers): the forward's arg-max table"""
    R, C = x.shape
    with torch.cuda.device(x.device):
        dgb = torch.empty(2, C, dtype=torch.float32, device=x.device)
        part = torch.empty(_lib.bq_bn_chunks(R, int(S), int(bool(pool))) * 2 * C, dtype=torch.float32, device=x.device)
        if pool and arg is not None and C <= 256 and 256 % C == 0:
            _check(_lib.bq_bn_backward_reduce_arg(_p(dy), _p(x), _p(arg), _p(stats[0]), _p(stats[1]), _p(stats[2]), _p(stats[3]),
                                                  _p(part), _p(dgb), R, C, int(S), int(bool(relu)), _stream()),
                   "bn_backward_reduce_arg")
        else:
            _check(_lib.bq_bn_backward_reduce(_p(dy), _p(x), _p(stats[0]), _p(stats[1]), _p(stats[2]), _p(stats[3]), _p(part),
                                              _p(dgb), R, C, int(S), int(bool(relu)), int(bool(pool)), _stream()),
                   "bn_backward_reduce")
    return dgb


def sa_bwd_fused(x, y_raw, dout, arg, w_pad, stats, dgb, S, relu, pool, need_dx, x_stats=None, carry_reduce=False):
    """x bf16 (R, ldx) whole padded rows, y_raw bf16 (R, N), dout bf16 (R, N) | (R // S, N) with arg u8 when pool, w_pad bf16
    (N, Kc), stats f32 (>= 4, N), dgb f32 (2, N) -> dx bf16 (R, ldx) | None, dw f32 (N, ldx).  x_stats: x is the previous
    layer's stored pre-activation and the layer's input relu(x * x_stats[0] + x_stats[1]) (bq_sa_bwd_fused_x; dx required);
    carry_reduce (with x_stats): a third value, f32 (2, ldx) = the PREVIOUS layer's dbeta | dgamma summed in the same pass
    (bq_sa_bwd_fused_xr), or None where the shape has no room for it"""
    R, ldx = x.shape[0], x.stride(0)
    N = y_raw.shape[1]
    with torch.cuda.device(x.device):
        dx = torch.empty(R, ldx, dtype=torch.bfloat16, device=x.device) if need_dx else None
        dw = torch.empty(N, ldx, dtype=torch.float32, device=x.device)
        wgs = _lib.bq_sa_bwd_workgroups(R, ldx, N, int(bool(pool)), int(bool(need_dx)))
        part = torch.empty(wgs * N * ldx, dtype=torch.float32, device=x.device)
        carry = bool(carry_reduce and x_stats is not None and need_dx
                     and _lib.bq_sa_bwd_reduce_supported(ldx, N, int(S), int(bool(pool))))
        red_part = torch.empty(wgs * 4 * 2 * ldx, dtype=torch.float32, device=x.device) if carry else None
        red_dgb = torch.empty(2, ldx, dtype=torch.float32, device=x.device) if carry else None
        xs = x_stats
        _check(_lib.bq_sa_bwd_fused_xr(_p(x), _p(xs[0]) if xs is not None else None, _p(xs[1]) if xs is not None else None,
                                       _p(xs[2]) if carry else None, _p(xs[3]) if carry else None, _p(red_part), _p(red_dgb),
                                       _p(y_raw), _p(dout), _p(arg), _p(w_pad), _p(stats[0]), _p(stats[1]), _p(stats[2]),
                                       _p(stats[3]), _p(dgb), _p(dx), _p(dw), _p(part), R, ldx, N, w_pad.stride(0), ldx, int(S),
                                       int(bool(relu)), int(bool(pool)), _stream()), "sa_bwd_fused")
    return (dx, dw, red_dgb) if carry_reduce else (dx, dw)


# SharedMLP outputs from the fp32 accumulators (bq_pwconv_bn_apply) instead of from the stored bf16 y.  OFF: measured in round 5
#  -- the layer output's error drops to one bf16 rounding, the 200-step convergence gap to fp32 does NOT
# close (+7.2 % against +3.3 % with the stored pre-activation, fp32's own spread 13 %) and the c3 step costs 1.3 ms more
FP32_PREACT = [False]
_lib.bq_pwconv_bn_apply.argtypes = [_vp, _l, _i, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _vp]
_lib.bq_pwconv_bn_apply.restype = ctypes.c_int


# ---- LM head + label-smoothed cross entropy (csrc/lmhead.hip + the cross-entropy epilogue of csrc/gemm.hip) -----------
_lib.bq_lmhead_ce_combine.argtypes = [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp]
_lib.bq_lmhead_ce_combine.restype = ctypes.c_int
_lib.bq_lmhead_ce_dlogits.argtypes = [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp]
_lib.bq_lmhead_ce_dlogits.restype = ctypes.c_int


def lmhead_ce_fwd(h, w, bias_pad, target, V, label_smoothing):
    """h bf16 (R, D), w bf16 (V, D) (contiguous), bias_pad f32 (Vp,) with Vp = V rounded up to 64 and zeros beyond V,
    target int32 (R,) (< 0: ignored).  Returns logits bf16 (R, Vp) (columns >= V are padding), loss f32 (R,), lse f32 (R,)."""
    R, D = h.shape
    Vp = bias_pad.numel()
    tiles_i = (Vp + 255) // 256
    with torch.cuda.device(h.device):
        logits = torch.empty(R, Vp, dtype=torch.bfloat16, device=h.device)
        part = torch.empty(tiles_i * 2, R, 3, dtype=torch.float32, device=h.device)
        zt = torch.zeros(R, dtype=torch.float32, device=h.device)
        loss = torch.empty(R, dtype=torch.float32, device=h.device)
        lse = torch.empty(R, dtype=torch.float32, device=h.device)
        # P = the vocabulary matrix with its TRUE extent in p_bytes (rows V .. Vp-1 of the last tile are out of bounds
        # for the bounds-checked LDS-DMA: zeros), Ni = the padded width of the logits
        gemm_grouped([dict(P=w, Q=h, out=logits, bias=bias_pad, out2=part, aux=target, colsum=zt, ksplit=V, Ni=Vp,
                           p_bytes=w.shape[0] * w.stride(0) * 2)], 0, EPI_BIAS_CE, 256)
        _check(_lib.bq_lmhead_ce_combine(_p(part), _p(zt), _p(target), _p(loss), _p(lse), R, tiles_i * 2, int(V),
                                         float(label_smoothing), _stream()), "lmhead_ce_combine")
    return logits, loss, lse


def lmhead_ce_dlogits(logits, lse, target, grad_loss, V, label_smoothing):
    """logits (R, Vp) bf16 from lmhead_ce_fwd -> dlogits IN PLACE (returned)"""
    R, Vp = logits.shape
    with torch.cuda.device(logits.device):
        _check(_lib.bq_lmhead_ce_dlogits(_p(logits), _p(lse), _p(target), _p(grad_loss), R, int(V), logits.stride(0),
                                         float(label_smoothing), _stream()), "lmhead_ce_dlogits")
    return logits


# ---- detection loss + gradient in two launches (csrc/detloss.hip) ---------------------------------------------------------
_DET_SCORES = ("objectness_scores", "heading_scores", "heading_residuals_normalized", "size_scores",
               "size_residuals_normalized", "sem_cls_scores")
_DET_TERM = {"vote_xyz": 0, "objectness_scores": 1, "center": 2, "heading_scores": 3, "heading_residuals_normalized": 4,
             "size_scores": 5, "size_residuals_normalized": 6, "sem_cls_scores": 7}


class _DetLossDesc(ctypes.Structure):
    _fields_ = ([(n, _vp) for n in (
        "seed_xyz", "vote_xyz", "aggregated_vote_xyz", "objectness_scores", "center", "heading_scores",
        "heading_residuals_normalized", "size_scores", "size_residuals_normalized", "sem_cls_scores", "seed_inds", "vote_label",
        "vote_label_mask", "center_label", "box_label_mask", "heading_class_label", "size_class_label", "sem_cls_label",
        "heading_residual_label", "size_residual_label", "mean_size_arr", "terms", "objectness_label", "objectness_mask",
        "object_assignment", "g_vote_xyz", "g_objectness_scores", "g_center", "g_heading_scores",
        "g_heading_residuals_normalized", "g_size_scores", "g_size_residuals_normalized", "g_sem_cls_scores", "scratch")]
        + [(n, _i) for n in ("B", "S", "VF", "N", "K", "G", "NH", "NS", "NC", "cl_ld", "seed_inds_i64")]
        + [("ld_" + n, _i) for n in _DET_SCORES]
        + [(n, _f) for n in ("near_threshold", "far_threshold", "objectness_weight_neg", "objectness_weight_pos")])


class _DetLossSeg(ctypes.Structure):
    _fields_ = [("g", _vp), ("out", _vp), ("rows", _i), ("width", _i), ("ld", _i), ("term", _i)]


_lib.bq_det_loss_fwd.argtypes = [ctypes.POINTER(_DetLossDesc), _vp]
_lib.bq_det_loss_fwd.restype = ctypes.c_int
_lib.bq_det_loss_bwd.argtypes = [ctypes.POINTER(_DetLossSeg), _i, _vp, _vp]
_lib.bq_det_loss_bwd.restype = ctypes.c_int


def det_score_ld(v, K):
    """row stride (floats per proposal) of a score tensor (B, K, w) or (B, K, NS, 3) the kernel can read in place: unit stride
    inside a row, rows `ld` apart, batches K * ld apart (a contiguous tensor, or a channel slice of the head output); else None"""
    w = v.shape[2] * (v.shape[3] if v.dim() == 4 else 1)
    ld = v.stride(1)
    if v.stride(-1) != 1 or ld < w or v.stride(0) != K * ld or (v.dim() == 4 and v.stride(2) != v.shape[3]):
        return None
    return ld


def det_loss_packing(t):
    """The six score tensors of the proposal head are slices of ONE (B, K, channels) tensor (proposal_module.decode_scores):
    -> (base, {name: (channel offset, width)}, channels) when they all are views, INSIDE the autograd graph, of the same
    contiguous base with one row stride -- the gradient is then produced for the base in one buffer; else None (the tensors
    are differentiated one by one, e.g. graphed.wrap_loss' fresh leaves)"""
    base = getattr(t[_DET_SCORES[0]], "_base", None)
    if base is None or not base.is_contiguous() or base.dtype != torch.float32 or not base.requires_grad:
        return None
    B, K = t["center"].shape[:2]
    if base.numel() % (B * K):
        return None
    ld = base.numel() // (B * K)
    off = {}
    for n in _DET_SCORES:
        v = t[n]
        if getattr(v, "_base", None) is not base or v.grad_fn is None or det_score_ld(v, K) != ld:
            return None
        o = v.storage_offset() - base.storage_offset()
        width = v.shape[2] * (v.shape[3] if v.dim() == 4 else 1)
        if o < 0 or o + width > ld:
            return None
        off[n] = (o, width)
    return base, off, ld


def det_loss_fwd(t, mean_size, near, far, w_neg, w_pos, packing=None):
    """t: dict of the tensors named in bq_det_loss_desc (fp32 outputs / labels, int64 class labels and masks; seed_inds int32
    or int64).  packing: det_loss_packing(t) -- the score tensors as slices of one head output: their gradients then land in
    ONE buffer shaped like it.  -> terms f32 (16,), objectness_label i64 (B,K), objectness_mask f32 (B,K), object_assignment
    i64 (B,K), grads {name or "packed": tensor}"""
    dev = t["center"].device
    B, K = t["center"].shape[:2]
    S = t["seed_xyz"].shape[1]
    VF = t["vote_xyz"].shape[1] // S
    N, G = t["vote_label"].shape[1], t["center_label"].shape[1]
    NH, NS, NC = t["heading_scores"].shape[2], t["size_scores"].shape[2], t["sem_cls_scores"].shape[2]
    d = _DetLossDesc()
    plain = ["seed_xyz", "vote_xyz", "aggregated_vote_xyz", "center", "vote_label", "center_label", "box_label_mask",
             "heading_residual_label", "size_residual_label"]
    for n in plain:
        _req(t[n], torch.float32, n)
        setattr(d, n, _p(t[n]))
    for n in ("vote_label_mask", "heading_class_label", "size_class_label", "sem_cls_label"):
        if t[n].dtype != torch.int64 or not t[n].is_contiguous() or not t[n].is_cuda:
            raise RuntimeError("%s must be a contiguous int64 device tensor" % n)
        setattr(d, n, _p(t[n]))
    si = t["seed_inds"]
    if si.dtype not in (torch.int32, torch.int64) or not si.is_contiguous():
        raise RuntimeError("seed_inds must be a contiguous int32 / int64 tensor")
    d.seed_inds, d.seed_inds_i64 = _p(si), int(si.dtype == torch.int64)
    _req(mean_size, torch.float32, "mean_size_arr")
    d.mean_size_arr = _p(mean_size)
    with torch.cuda.device(dev):
        terms = torch.zeros(16, dtype=torch.float32, device=dev)
        lab = torch.empty(B, K, dtype=torch.int64, device=dev)
        msk = torch.empty(B, K, dtype=torch.float32, device=dev)
        asg = torch.empty(B, K, dtype=torch.int64, device=dev)
        scratch = torch.empty(B * G + 1024, dtype=torch.int32, device=dev)
        grads = {"vote_xyz": torch.empty_like(t["vote_xyz"]), "center": torch.empty_like(t["center"])}
        d.g_vote_xyz, d.g_center = _p(grads["vote_xyz"]), _p(grads["center"])
        if packing:
            base, off, ld = packing
            gp = grads["packed"] = torch.empty_like(base)
            for n in _DET_SCORES:
                setattr(d, n, t[n].data_ptr())
                setattr(d, "g_" + n, gp.data_ptr() + 4 * off[n][0])
                setattr(d, "ld_" + n, ld)
        else:
            for n in _DET_SCORES:
                v = t[n]
                ld = det_score_ld(v, K)
                if ld is None or v.dtype != torch.float32 or not v.is_cuda:
                    raise RuntimeError("%s: fp32 device rows with unit stride expected" % n)
                grads[n] = torch.empty_strided(v.shape, v.stride(), dtype=torch.float32, device=dev)   # (same row stride)
                setattr(d, n, v.data_ptr())
                setattr(d, "g_" + n, grads[n].data_ptr())
                setattr(d, "ld_" + n, ld)
        d.terms, d.objectness_label, d.objectness_mask, d.object_assignment, d.scratch = _p(terms), _p(lab), _p(msk), _p(asg), _p(scratch)
        d.B, d.S, d.VF, d.N, d.K, d.G, d.NH, d.NS, d.NC = B, S, VF, N, K, G, NH, NS, NC
        d.cl_ld = t["center_label"].shape[2]
        d.near_threshold, d.far_threshold, d.objectness_weight_neg, d.objectness_weight_pos = near, far, w_neg, w_pos
        _check(_lib.bq_det_loss_fwd(ctypes.byref(d), _stream()), "det_loss_fwd")
    return terms, lab, msk, asg, grads


def det_loss_bwd(grads, upstream, packing=None):
    """grads: det_loss_fwd's buffers, upstream f32 (8,) -> {name or "packed": g * upstream[term]}, ONE launch"""
    outs = {k: torch.empty_strided(g.shape, g.stride(), dtype=g.dtype, device=g.device) for k, g in grads.items()}
    segs = []

    def seg(g, o, g_off, rows, width, ld, term):
        segs.append(_DetLossSeg(g.data_ptr() + 4 * g_off, o.data_ptr() + 4 * g_off, rows, width, ld, term))
    for n in ("vote_xyz", "center"):
        seg(grads[n], outs[n], 0, grads[n].numel() // 3, 3, 3, _DET_TERM[n])
    if packing:
        base, off, ld = packing
        rows = base.numel() // ld
        covered = sorted((o, w, n) for n, (o, w) in off.items())
        pos = 0
        for o, w, n in covered:
            if o > pos:
                seg(grads["packed"], outs["packed"], pos, rows, o - pos, ld, -1)   # channels no term reads (the centre offsets)
            seg(grads["packed"], outs["packed"], o, rows, w, ld, _DET_TERM[n])
            pos = o + w
        if pos < ld:
            seg(grads["packed"], outs["packed"], pos, rows, ld - pos, ld, -1)
    else:
        for n in _DET_SCORES:
            g = grads[n]
            w = g.shape[2] * (g.shape[3] if g.dim() == 4 else 1)
            seg(g, outs[n], 0, g.shape[0] * g.shape[1], w, g.stride(1), _DET_TERM[n])
    arr = (_DetLossSeg * len(segs))(*segs)
    with torch.cuda.device(upstream.device):
        _check(_lib.bq_det_loss_bwd(arr, len(segs), _p(upstream), _stream()), "det_loss_bwd")
    return outs


# ---- deterministic scatter gradients through an inverted index (csrc/invert.hip) ---------------------------------------------
DETERMINISTIC_SCATTER = [True]
_lib.bq_invert_index_workspace_bytes.argtypes = [_l]
_lib.bq_invert_index_workspace_bytes.restype = ctypes.c_size_t
_lib.bq_invert_index.argtypes = [_vp, _i, _l, _i, _vp, _vp, _vp, ctypes.c_size_t, _vp]
_lib.bq_invert_index.restype = ctypes.c_int
_lib.bq_group_concat_pm_grad_gather.argtypes = [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _vp]
_lib.bq_group_concat_pm_grad_gather.restype = ctypes.c_int
_lib.bq_three_interpolate_grad_gather.argtypes = [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]
_lib.bq_three_interpolate_grad_gather.restype = ctypes.c_int


def invert_index(idx, N):
    """idx int32 (B, ...) with values in [0, N) -> (start int32 (B * N + 1,), slots int32-sized uint32 (B * L,)): for scene b
    and value v the positions b * L + l with idx[b].flatten()[l] == v are slots[start[b * N + v] : start[b * N + v + 1]],
    ascending"""
    _req(idx, torch.int32, "idx")
    B = idx.shape[0]
    L = idx.numel() // max(B, 1)
    with torch.cuda.device(idx.device):
        start = torch.empty(B * int(N) + 1, dtype=torch.int32, device=idx.device)
        slots = torch.empty(max(B * L, 1), dtype=torch.int32, device=idx.device)
        nbytes = int(_lib.bq_invert_index_workspace_bytes(B * L))
        ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=idx.device)
        _check(_lib.bq_invert_index(_p(idx), B, L, int(N), _p(start), _p(slots), _p(ws), nbytes, _stream()), "invert_index")
    return start, slots


def group_concat_pm_grad_gather(grad_out, inv, n, radius=1.0, normalize=False, need_features=True, need_xyz=False,
                                need_new_xyz=False):
    """the gradients of group_concat_pm as gathers over inv = invert_index(idx, n): grad_out (B, M, S, 3 + C) rows (contiguous
    or padded rows of a uniform stride % 8 == 0) -> (grad_feats f32 (B, n, C) | None, grad_xyz f32 (B, n, 3) | None,
    grad_new_xyz f32 (B, M, 3) | None); None (a single value) when the layout is not the kernel's"""
    B, M, S, CT = grad_out.shape
    C = CT - 3
    ld = grad_out.stride(2)
    if (grad_out.stride(3) != 1 or grad_out.stride(1) != S * ld or grad_out.stride(0) != M * S * ld or ld < CT or ld % 8
            or grad_out.data_ptr() % 16 or grad_out.dtype not in (torch.bfloat16, torch.float32)):
        return None
    need_features = bool(need_features) and C > 0
    if not (need_features or need_xyz or need_new_xyz):
        return None, None, None
    dev = grad_out.device
    with torch.cuda.device(dev):
        gf = torch.empty(B, int(n), C, dtype=torch.float32, device=dev) if need_features else None
        gx = torch.empty(B, int(n), 3, dtype=torch.float32, device=dev) if need_xyz else None
        gn = torch.empty(B, M, 3, dtype=torch.float32, device=dev) if need_new_xyz else None
        _check(_lib.bq_group_concat_pm_grad_gather(_p(grad_out), int(grad_out.dtype == torch.bfloat16), _p(inv[0]), _p(inv[1]),
                                                   _p(gf), _p(gx), _p(gn), B, C, int(n), M, S, ld, float(radius),
                                                   int(bool(normalize)), _stream()), "group_concat_pm_grad_gather")
    return gf, gx, gn


def three_interpolate_grad_gather(grad_out, inv, weight, m):
    """grad_out f32 (B, C, n) contiguous, inv = invert_index(idx (B, n, 3), m), weight f32 (B, n, 3) -> f32 (B, C, m)"""
    _req(grad_out, torch.float32, "grad_out"); _req(weight, torch.float32, "weight")
    B, C, n = grad_out.shape
    with torch.cuda.device(grad_out.device):
        out = torch.empty(B, C, int(m), dtype=torch.float32, device=grad_out.device)
        _check(_lib.bq_three_interpolate_grad_gather(_p(grad_out), _p(inv[0]), _p(inv[1]), _p(weight), _p(out), B, C, n, int(m),
                                                     _stream()), "three_interpolate_grad_gather")
    return out

"""ctypes front-end of oracle/libpn2oracle.so presenting the reference's `pointnet2._ext`
surface (lib/pointnet2/_ext_src/src/bindings.cpp:6-19) on CPU torch tensors.

TEST INFRASTRUCTURE ONLY (see pointnet2_oracle.c header).  Allocation / zero-fill / dtype and
contiguity checks mirror the reference's C++ wrappers (sampling.cpp, ball_query.cpp,
group_points.cpp, interpolate.cpp); the only difference is that CPU tensors are accepted --
the reference asserts "CPU not supported".
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpn2oracle.so")
_lib = None


def build(force=False):
    """Compile the oracle with gcc (recipe: oracle/Makefile)."""
    src = os.path.join(_HERE, "pointnet2_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libpn2oracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.oracle_opt_n_threads.restype = ctypes.c_int
        _lib.oracle_opt_n_threads.argtypes = [ctypes.c_int]
    return _lib


class FmaVariant:
    """The index-producing operators of the SAME C file built with its squared distances FMA-contracted (`ORACLE_FMA` = 1 or 2,
    oracle/Makefile).  Sensitivity study only (tests/test_oracle.py::test_fma_contraction_sensitivity): how many indices move
    if the reference's nvcc build contracted `a*a + b*b + c*c` (sampling_gpu.cu:97-107 under the default -fmad=true)."""

    def __init__(self, variant):
        assert variant in (1, 2)
        path = os.path.join(_HERE, "libpn2oracle_fma%d.so" % variant)
        src = os.path.join(_HERE, "pointnet2_oracle.c")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            # -mfma (when the host has it) inlines fmaf; without it libm's correctly rounded fmaf is called: same results
            has_fma = " fma " in open("/proc/cpuinfo").read() if os.path.exists("/proc/cpuinfo") else False
            subprocess.check_call(["make", "-C", _HERE, "-s", "-B", os.path.basename(path)] +
                                  (["CFLAGS_EXTRA=-mfma"] if has_fma else []))
        self._l = ctypes.CDLL(path)
        self._l.oracle_fma_variant.restype = ctypes.c_int
        assert self._l.oracle_fma_variant() == variant

    def furthest_point_sampling(self, points, nsamples):
        return furthest_point_sampling(points, nsamples, _l=self._l)

    def ball_query(self, new_xyz, xyz, radius, nsample):
        return ball_query(new_xyz, xyz, radius, nsample, _l=self._l)

    def three_nn(self, unknowns, knows):
        return three_nn(unknowns, knows, _l=self._l)


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _chk(t, dtype, name):
    if t.is_cuda:
        raise RuntimeError("oracle is CPU-only: %s is a device tensor" % name)
    if not t.is_contiguous():
        raise RuntimeError("%s must be a contiguous tensor" % name)
    if t.dtype != dtype:
        raise RuntimeError("%s must be a %s tensor" % (name, dtype))


def opt_n_threads(n):
    return lib().oracle_opt_n_threads(int(n))


def furthest_point_sampling(points, nsamples, _l=None):
    _chk(points, torch.float32, "points")
    B, N, _ = points.shape
    out = torch.zeros(B, nsamples, dtype=torch.int32)
    tmp = torch.full((B, N), 1e10, dtype=torch.float32)
    (_l or lib()).oracle_furthest_point_sampling(B, N, int(nsamples), _p(points), _p(tmp), _p(out))
    return out


def gather_points(points, idx):
    _chk(points, torch.float32, "points"); _chk(idx, torch.int32, "idx")
    B, C, N = points.shape
    M = idx.shape[1]
    out = torch.zeros(B, C, M, dtype=torch.float32)
    lib().oracle_gather_points(B, C, N, M, _p(points), _p(idx), _p(out))
    return out


def gather_points_grad(grad_out, idx, n):
    _chk(grad_out, torch.float32, "grad_out"); _chk(idx, torch.int32, "idx")
    B, C, M = grad_out.shape
    out = torch.zeros(B, C, n, dtype=torch.float32)
    lib().oracle_gather_points_grad(B, C, int(n), M, _p(grad_out), _p(idx), _p(out))
    return out


def ball_query(new_xyz, xyz, radius, nsample, _l=None):
    _chk(new_xyz, torch.float32, "new_xyz"); _chk(xyz, torch.float32, "xyz")
    B, M, _ = new_xyz.shape
    N = xyz.shape[1]
    idx = torch.zeros(B, M, nsample, dtype=torch.int32)
    (_l or lib()).oracle_ball_query(B, N, M, ctypes.c_float(radius), int(nsample), _p(new_xyz), _p(xyz), _p(idx))
    return idx


def group_points(points, idx):
    _chk(points, torch.float32, "points"); _chk(idx, torch.int32, "idx")
    B, C, N = points.shape
    _, M, S = idx.shape
    out = torch.zeros(B, C, M, S, dtype=torch.float32)
    lib().oracle_group_points(B, C, N, M, S, _p(points), _p(idx), _p(out))
    return out


def group_points_grad(grad_out, idx, n):
    _chk(grad_out, torch.float32, "grad_out"); _chk(idx, torch.int32, "idx")
    B, C, M, S = grad_out.shape
    out = torch.zeros(B, C, n, dtype=torch.float32)
    lib().oracle_group_points_grad(B, C, int(n), M, S, _p(grad_out), _p(idx), _p(out))
    return out


def three_nn(unknowns, knows, _l=None):
    _chk(unknowns, torch.float32, "unknowns"); _chk(knows, torch.float32, "knows")
    B, n, _ = unknowns.shape
    m = knows.shape[1]
    idx = torch.zeros(B, n, 3, dtype=torch.int32)
    dist2 = torch.zeros(B, n, 3, dtype=torch.float32)
    (_l or lib()).oracle_three_nn(B, n, m, _p(unknowns), _p(knows), _p(dist2), _p(idx))
    return [dist2, idx]


def three_interpolate(points, idx, weight):
    _chk(points, torch.float32, "points"); _chk(idx, torch.int32, "idx"); _chk(weight, torch.float32, "weight")
    B, C, m = points.shape
    n = idx.shape[1]
    out = torch.zeros(B, C, n, dtype=torch.float32)
    lib().oracle_three_interpolate(B, C, m, n, _p(points), _p(idx), _p(weight), _p(out))
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    _chk(grad_out, torch.float32, "grad_out"); _chk(idx, torch.int32, "idx"); _chk(weight, torch.float32, "weight")
    B, C, n = grad_out.shape
    out = torch.zeros(B, C, m, dtype=torch.float32)
    lib().oracle_three_interpolate_grad(B, C, n, int(m), _p(grad_out), _p(idx), _p(weight), _p(out))
    return out

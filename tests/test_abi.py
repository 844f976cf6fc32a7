"""The C-ABI library loads and exports every symbol include/bqhip.h declares; host-side argument
checks behave like the reference's wrappers.  No GPU, no compute."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT


def declared_symbols():
    hdr = "".join(open(os.path.join(ROOT, "include", h)).read() for h in ("bqhip.h", "bqhip_fusion.h"))
    return sorted(set(re.findall(r"BQ_API\s+(?:const\s+)?\w+\s*\*?\s*(bq_\w+)\s*\(", hdr)))


def test_headers_compile_as_c_and_cover_the_fusion_kernels():
    import subprocess, tempfile
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.c")
        open(src, "w").write('#include "bqhip_fusion.h"\nint main(void) { return BQHIP_ABI_VERSION - 3; }\n')
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", src,
                               "-o", os.path.join(d, "t.o")])
    syms = declared_symbols()
    for s in ("bq_attn_fwd", "bq_attn_bwd", "bq_drop_add_ln_fwd", "bq_drop_add_ln_bwd", "bq_colsum_bf16",
              "bq_gemm_bf16", "bq_colsum_grouped_bf16", "bq_pwconv_bn_fwd", "bq_group_concat_pm",
              "bq_group_concat_pm_grad"):
        assert s in syms


def test_header_declares_the_nine_reference_operators():
    syms = declared_symbols()
    for op in ("furthest_point_sampling", "gather_points", "gather_points_grad", "ball_query", "group_points",
               "group_points_grad", "three_nn", "three_interpolate", "three_interpolate_grad"):
        assert "bq_" + op in syms  # bindings.cpp:6-19


def test_library_exports_every_declared_symbol():
    from bridgeqa_amd import _ext
    lib = ctypes.CDLL(_ext.library_path())
    for s in declared_symbols():
        assert hasattr(lib, s), s
    assert lib.bq_abi_version() == 3


def test_shim_surface_and_cpu_rejection():
    from bridgeqa_amd import _ext
    for op in ("furthest_point_sampling", "gather_points", "gather_points_grad", "ball_query", "group_points",
               "group_points_grad", "three_nn", "three_interpolate", "three_interpolate_grad"):
        assert callable(getattr(_ext, op))
    with pytest.raises(RuntimeError, match="CPU not supported"):
        _ext.furthest_point_sampling(torch.rand(1, 8, 3), 4)
    with pytest.raises(RuntimeError, match="CPU not supported"):
        _ext.ball_query(torch.rand(1, 2, 3), torch.rand(1, 8, 3), 0.5, 4)


def test_bad_extents_return_status_not_exit():
    from bridgeqa_amd import _ext
    lib = _ext._lib
    # negative extents / null pointers are rejected before any launch
    assert lib.bq_furthest_point_sampling(None, None, 0, None, 1, 0, 4, None) == -1
    assert lib.bq_furthest_point_sampling(None, None, 0, None, 1, 1 << 23, 4, None) == -2
    assert b"fps" in lib.bq_last_error()
    assert lib.bq_ball_query(None, None, None, -1, 4, 4, 0.5, 4, None) == -1
    assert lib.bq_group_concat(None, None, None, None, None, 1, 0, 4, 4, 6, 0.5, 1, None) == -1
    # empty work is a no-op success, as in the reference (`if (m <= 0) return`)
    assert lib.bq_furthest_point_sampling(None, None, 0, None, 0, 8, 4, None) == 0
    assert lib.bq_fps_workspace_bytes(16, 40000) == 16 * 40000 * 20 and lib.bq_fps_workspace_bytes(16, 2048) == 0
    assert lib.bq_opt_n_threads(40000) == 512 and lib.bq_opt_n_threads(300) == 256

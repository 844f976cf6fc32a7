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
        open(src, "w").write('#include "bqhip_fusion.h"\nint main(void) { return BQHIP_ABI_VERSION - 6; }\n')
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
    assert lib.bq_abi_version() == 6


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


def test_transpose_table_packing_matches_the_header_record():
    """_ext.transpose_table: 32-byte records {src, dst, N, K, ld, tiles_k} as bq_transpose_multi_bf16 reads them, one
    {tensor, tile} chunk per 64 x 64 tile; unsupported pairs are refused on the host (no GPU needed: pointers only)"""
    import struct
    import pytest
    import torch
    from bridgeqa_amd import _ext
    assert _ext.TRANSPOSE_TENSOR_BYTES == 32
    cat = torch.zeros(256, 128, dtype=torch.bfloat16)
    pairs = [(torch.zeros(128, 192, dtype=torch.bfloat16), torch.zeros(192, 128, dtype=torch.bfloat16)),
             (cat[64:192], torch.zeros(128, 128, dtype=torch.bfloat16)),
             (torch.zeros(64, 320, dtype=torch.bfloat16)[:, :256], torch.zeros(256, 64, dtype=torch.bfloat16))]
    table, chunks = _ext.transpose_table(pairs, torch.device("cpu"))
    raw = bytes(table.numpy().tobytes())
    assert len(raw) == 32 * len(pairs)
    for i, (src, dst) in enumerate(pairs):
        p_src, p_dst, N, K, ld, tk = struct.unpack("<QQiiii", raw[32 * i:32 * (i + 1)])
        assert (p_src, p_dst, N, K, ld, tk) == (src.data_ptr(), dst.data_ptr(), src.shape[0], src.shape[1], src.stride(0), src.shape[1] // 64)
    want = [(i, t) for i, (s, _) in enumerate(pairs) for t in range(s.shape[0] // 64 * (s.shape[1] // 64))]
    assert [tuple(r) for r in chunks.tolist()] == want and chunks.dtype == torch.int32
    for bad in ((torch.zeros(100, 128, dtype=torch.bfloat16), torch.zeros(128, 100, dtype=torch.bfloat16)),     # N % 64
                (torch.zeros(128, 128), torch.zeros(128, 128)),                                                # dtype
                (torch.zeros(128, 128, dtype=torch.bfloat16), torch.zeros(128, 192, dtype=torch.bfloat16)),    # dst shape
                (torch.zeros(128, 256, dtype=torch.bfloat16)[:, ::2], torch.zeros(128, 128, dtype=torch.bfloat16))):  # stride
        with pytest.raises(RuntimeError):
            _ext.transpose_table([bad], torch.device("cpu"))

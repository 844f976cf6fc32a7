"""BLIP-ITM view ranking on the GPU: fp32 on the device, and bf16 through the HIP kernels (MFMA GEMMs, fused attention,
add + LayerNorm) against the reference's fp32 golden -- contrastive similarities are cosines of 256-d features, bf16
operands move them by < 1e-2."""
import pytest
import torch

from test_itm_cpu import run_itm

pytestmark = pytest.mark.gpu


def test_itm_fp32_on_device(dev):
    run_itm(dev, 1e-3, 1e-4)


def test_itm_bf16_hip_path(dev):
    from bridgeqa_amd import fusion_ops as ops
    prev = ops.set_compute_dtype(torch.bfloat16)
    try:
        run_itm(dev, 0.0, 1.5e-2)
    finally:
        ops.set_compute_dtype(prev)

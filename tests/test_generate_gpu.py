"""The beam search on the device: fp32 (cache path == no-cache path, exhaustive search) and the BLIP_VQA3D generate mode
under the bf16 kernels.  (Parity unpinned: bridgeqa_amd/generation.py.)"""
import pytest
import torch

from test_generate_cpu import run_blip_generate, run_exhaustive, run_greedy_equivalence

pytestmark = pytest.mark.gpu


def test_one_beam_is_greedy_decoding_on_device(dev):
    run_greedy_equivalence(dev)


def test_wide_beam_search_is_exhaustive_on_device(dev):
    run_exhaustive(dev, 1.0)


def test_blip_vqa3d_generate_mode_bf16(dev):
    from bridgeqa_amd import fusion_ops as ops
    prev = ops.set_compute_dtype(torch.bfloat16)
    try:
        run_blip_generate(dev)
    finally:
        ops.set_compute_dtype(prev)

"""bridgeqa_amd.enet on the device (fp32, the library's convolutions -- offline preparation, not the training path) against
the reference goldens, and the batched extraction entry point."""
import numpy as np
import pytest
import torch

from test_enet_cpu import GOLD, golden_frames, run_enet

pytestmark = pytest.mark.gpu


def test_enet_fp32_on_device_vs_reference_golden(dev):
    run_enet(dev, 2e-3, 2e-4)


def test_extract_features_batches_and_matches_a_single_pass(dev):
    from bridgeqa_amd import enet
    g = np.load(GOLD)
    frames, _ = golden_frames(g)
    net = enet.feature_extractor(None, device=dev)
    frames = torch.from_numpy(np.concatenate([frames, frames[::-1]], 0))
    a = enet.extract_features(net, frames, batch_size=3)
    b = net(enet.preprocess_frames(frames.to(dev)))
    assert tuple(a.shape) == (4, 128, 32, 41) and a.dtype == torch.float32 and a.is_cuda
    assert ((a - b).norm() / b.norm()).item() < 1e-4
    assert ((a[0] - a[3]).norm() / a[0].norm()).item() < 1e-4          # same frame in different batches
    prepared = enet.preprocess_frames(frames[:2].to(dev))
    assert torch.allclose(prepared.cpu(), enet.preprocess_frames(frames[:2]), rtol=0, atol=1e-6)   # same pixels picked

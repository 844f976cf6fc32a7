"""bridgeqa_amd.enet (the ENet 2D feature extractor of the offline multiview preparation) in fp32 against golden vectors
from the reference's own lib/enet.py (oracle/gen_golden_enet.py): state-dict key set, arithmetic of every block kind, the
frozen / trainable / classifier split and the loader's image preparation.  CPU; tests/test_enet_gpu.py runs it on the device."""
import os

import numpy as np
import torch

from golden_util import fill_params, subsample

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "enet.npz")


def close(a, b, rtol, atol):
    a = a.detach().float().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    np.testing.assert_allclose(subsample(a), np.asarray(b), rtol=rtol, atol=atol * max(1.0, float(np.abs(b).max())))


def golden_input(g):
    x = torch.randn(1, 3, 256, 328, generator=torch.Generator().manual_seed(31))
    assert abs(float(x.double().sum()) - float(g["x_sum"])) < 1e-6 and np.array_equal(x[0, :, 0, :8].numpy(), g["x_head"])
    return x


def golden_frames(g):
    rng = np.random.RandomState(5)
    frames = rng.randint(0, 256, size=(2, 240, 320, 3)).astype(np.uint8)
    odd = rng.randint(0, 256, size=(1, 300, 420, 3)).astype(np.uint8)
    assert [int(frames.astype(np.int64).sum()), int(odd.astype(np.int64).sum())] == g["frames_sum"].tolist()
    return frames, odd


def run_enet(dev, rtol, atol):
    from bridgeqa_amd import enet
    g = np.load(GOLD)
    net = enet.create_enet(41)
    assert [k for k, _ in fill_params(net, "enet.")] == list(g["enet_keys"])
    net = net.to(dev).eval()
    x = golden_input(g).to(dev)
    with torch.no_grad():
        h = x
        for i in range(len(net)):
            h = net[i](h)
            if i == 3:
                close(h, g["after_initial"], rtol, atol)
            if i == 8:
                close(h, g["after_stage1"], rtol, atol)
            if i == 25:
                assert list(h.shape) == g["features_shape"].tolist()
                close(h, g["features"], rtol, atol)
                feats = h
        close(h, g["scores"], rtol, atol)
        # the split the 3D pipeline uses (lib/enet.py:697-717): same modules, regrouped
        fixed, trainable, classifier = enet.create_enet_for_3d(41, None)
        assert len(fixed) == 18 and len(trainable) == 8 and len(classifier) == 1
        assert not any(p.requires_grad for p in fixed.parameters()) and all(p.requires_grad for p in trainable.parameters())
        whole = torch.nn.Sequential(*fixed, *trainable, *classifier)
        whole.load_state_dict(net.state_dict())
        got = torch.nn.Sequential(fixed, trainable).to(dev).eval()(x)
        assert torch.equal(got, feats)
    return enet, g


def test_enet_fp32_vs_reference_golden():
    run_enet(torch.device("cpu"), 2e-4, 2e-5)


def test_scaled_dropout_keeps_the_inference_factor_and_cancels_it_in_training():
    from bridgeqa_amd import enet
    d = enet.ScaledDropout2d(0.1)
    x = torch.ones(4, 64, 3, 3)
    assert torch.allclose(d.eval()(x), x * 0.9)
    y = d.train()(x)
    kept = y[y != 0]
    assert torch.allclose(kept, torch.ones_like(kept))      # (1 - p) * 1 / (1 - p)


def test_frame_preparation_matches_the_loader():
    from bridgeqa_amd import enet
    g = np.load(GOLD)
    frames, odd = golden_frames(g)
    got = enet.preprocess_frames(torch.from_numpy(frames))
    assert tuple(got.shape) == (2, 3, 256, 328)
    np.testing.assert_allclose(subsample(got.numpy()), g["frames_prepared"], rtol=0, atol=1e-6)
    got = enet.preprocess_frames(torch.from_numpy(odd))
    np.testing.assert_allclose(subsample(got.numpy()), g["frames_odd_prepared"], rtol=0, atol=1e-6)
    same = torch.from_numpy(np.random.RandomState(1).randint(0, 256, size=(1, 256, 328, 3)).astype(np.uint8))
    ref = (same.permute(0, 3, 1, 2).float() / 255 - torch.tensor(enet.MEAN).view(1, 3, 1, 1)) / torch.tensor(enet.STD).view(1, 3, 1, 1)
    assert torch.equal(enet.preprocess_frames(same), ref)


def test_create_enet_for_3d_loads_a_checkpoint_file(tmp_path):
    """lib/enet.py:697-699: the 2D-pretrained state dict is read from model_path into the flat 27-entry network before
    it is regrouped; the feature extractor built from it reproduces the source network's features"""
    from bridgeqa_amd import enet
    src = enet.create_enet(41)
    fill_params(src, "enet.")
    path = str(tmp_path / "scannetv2_enet.pth")
    torch.save(src.state_dict(), path)
    fixed, trainable, classifier = enet.create_enet_for_3d(41, path)
    x = torch.randn(1, 3, 64, 80, generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        want = torch.nn.Sequential(*[src[i] for i in range(26)]).eval()(x)
        got = torch.nn.Sequential(fixed, trainable).eval()(x)
        assert torch.equal(got, want) and tuple(got.shape) == (1, 128, 8, 10)
        assert torch.equal(classifier(got), src[26](want))
        net = enet.feature_extractor(path)
        assert torch.equal(net(x), want) and not any(p.requires_grad for p in net.parameters())

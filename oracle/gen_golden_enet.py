"""Golden vectors for the ENet 2D feature extractor of the offline multiview preparation, produced by
the REFERENCE's own lib/enet.py create_enet (:130-695) in this build container.

TEST INFRASTRUCTURE.  Usage:  python oracle/gen_golden_enet.py   -> tests/golden/enet.npz

Weights are not stored: both sides fill every state-dict entry from a generator seeded by its name
(tests/golden_util.fill_params), which also pins the state-dict key set.  Stored: the key list, the class scores of one seeded input image, the 128-channel feature map the 3D pipeline keeps (entries 0..25, scripts/compute_multiview_features.py:88-98) and
two intermediate maps.

The loader's image preparation (compute_multiview_features.py:58-78) needs torchvision, which this image lacks: its three
statements -- Resize([h, w'], NEAREST) of a PIL image, CenterCrop([h, w]), Normalize -- are restated here with PIL and numpy
(torchvision's CenterCrop offsets: int(round((size - crop) / 2.0))); that part of the fixture is PARITY UNPINNED.
"""
import math
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden", "enet.npz")
sys.path.insert(0, os.path.join(REPO, "tests"))
from golden_util import fill_params, subsample  # noqa: E402


def loader_image(image, new_image_dims):
    """compute_multiview_features.py:58-78 on one (H, W, 3) uint8 frame, torchvision's transforms spelled out"""
    from PIL import Image
    image_dims = [image.shape[1], image.shape[0]]
    if image_dims != new_image_dims:
        resize_width = int(math.floor(new_image_dims[1] * float(image_dims[0]) / float(image_dims[1])))
        pil = Image.fromarray(image).resize((resize_width, new_image_dims[1]), Image.NEAREST)
        left = int(round((resize_width - new_image_dims[0]) / 2.0))
        top = int(round((new_image_dims[1] - new_image_dims[1]) / 2.0))
        pil = pil.crop((left, top, left + new_image_dims[0], top + new_image_dims[1]))
        image = np.array(pil)
    x = np.transpose(image, [2, 0, 1]).astype(np.float32) / 255.0
    mean = np.array([0.496342, 0.466664, 0.440796], dtype=np.float32).reshape(3, 1, 1)
    std = np.array([0.277856, 0.28623, 0.291129], dtype=np.float32).reshape(3, 1, 1)
    return (x - mean) / std


def main():
    sys.path.insert(0, REF)
    from lib.enet import create_enet
    torch.manual_seed(0)
    net = create_enet(41)
    keys = [k for k, _ in fill_params(net, "enet.")]
    net.eval()
    g = torch.Generator().manual_seed(31)
    x = torch.randn(1, 3, 256, 328, generator=g)          # (the test redraws it from the same seed; x_sum guards the draw)
    save = {"enet_keys": np.array(keys), "x_sum": np.array(float(x.double().sum())), "x_head": x[0, :, 0, :8].numpy()}
    with torch.no_grad():
        h = x
        for i in range(len(net)):
            h = net[i](h)
            if i == 3:
                save["after_initial"] = subsample(h.numpy())
            if i == 8:
                save["after_stage1"] = subsample(h.numpy())
            if i == 25:
                save["features"] = subsample(h.numpy())
                save["features_shape"] = np.array(h.shape)
        save["scores"] = subsample(h.numpy())
    rng = np.random.RandomState(5)
    frames = rng.randint(0, 256, size=(2, 240, 320, 3)).astype(np.uint8)       # ScanNet frame size
    save["frames_prepared"] = subsample(np.stack([loader_image(f, [328, 256]) for f in frames]))   # frames: RandomState(5)
    odd = rng.randint(0, 256, size=(1, 300, 420, 3)).astype(np.uint8)
    save["frames_odd_prepared"] = subsample(np.stack([loader_image(f, [328, 256]) for f in odd]))
    save["frames_sum"] = np.array([int(frames.astype(np.int64).sum()), int(odd.astype(np.int64).sum())])
    np.savez_compressed(OUT, **save)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", len(keys), "state-dict entries; features", save["features"].shape,
          "abs mean", float(np.abs(save["features"]).mean()))


if __name__ == "__main__":
    main()

"""The 2D feature extractor of the offline multiview preparation: the ENet encoder whose 128-channel
1/8-resolution maps `scripts/compute_multiview_features.py:88-122` stores per frame and `bridgeqa_amd.projection` then lifts
onto the points.  Mirrors `lib/enet.py` (create_enet :130-695, create_enet_for_3d :697-717) in state-dict layout -- the
reference is a Lua-Torch export, a flat nn.Sequential of 27 entries with numeric keys ("4.0.0.3.weight", ...), so that
`scannetv2_enet.pth` loads unchanged -- and in arithmetic (eval mode; its Dropout2d variants scale by (1 - p) at inference,
`lib/enet.py:89-95`).

Built from a table of the 22 bottlenecks instead of the export's unrolled listing.  Offline, inference-only, fp32 like the
reference: the convolutions are the library's (16-32 channel 3x3 / dilated / 5x1 kernels over 256 x 328 pixels, a one-time
cost per dataset; none of it is on the training path and no HIP kernel of this repo is involved).  `preprocess_frames`
restates the loader's resize / crop / normalise (`compute_multiview_features.py:58-78`) on the device.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

IMAGE_DIMS = (328, 256)                       # (width, height) the frames are brought to, compute_multiview_features.py:38
MEAN = (0.496342, 0.466664, 0.440796)         # :72
STD = (0.277856, 0.28623, 0.291129)


class Branches(nn.Sequential):
    """every child applied to the same input -> list (the export's ConcatTable)"""

    def forward(self, x):
        return [m(x) for m in self]


class Join(nn.Module):
    def __init__(self, how):
        super().__init__()
        self.how = how

    def forward(self, xs):
        return torch.cat(xs, 1) if self.how == "cat" else xs[0] + xs[1]


class Same(nn.Module):
    def forward(self, x):
        return x


class ChannelPad(nn.Module):
    """`extra` zero channels appended (the skip branch of a down-sampling bottleneck)"""

    def __init__(self, extra):
        super().__init__()
        self.extra = extra

    def forward(self, x):
        return F.pad(x, (0, 0, 0, 0, 0, self.extra))


class ScaledDropout2d(nn.Module):
    """the export's Dropout2d: the input is scaled by (1 - p) and THEN handed to nn.Dropout2d, which cancels the
    rescaling in training and leaves the factor in place at inference (lib/enet.py:89-95)"""

    def __init__(self, p):
        super().__init__()
        self.p = p

    def forward(self, x):
        return F.dropout2d(x * (1 - self.p), self.p, self.training)


def _bn(c):
    return nn.BatchNorm2d(c, 0.001, 0.1, True)


def _bottleneck(cin, cout, kind, arg, p_drop):
    """kind: "down" (2x2/2 projection + max-pooled, zero-padded skip), "plain", "dilated" (arg = rate), "asym" (1x5, 5x1)"""
    mid = cout // 4
    if kind == "down":
        first = nn.Conv2d(cin, mid, 2, 2, 0, bias=False)
    else:
        first = nn.Conv2d(cin, mid, 1, 1, 0, bias=False)
    main = [first, _bn(mid), nn.PReLU(mid)]
    if kind == "asym":
        main += [nn.Conv2d(mid, mid, (1, 5), 1, (0, 2), bias=False), nn.Conv2d(mid, mid, (5, 1), 1, (2, 0))]
    else:
        d = arg if kind == "dilated" else 1
        main += [nn.Conv2d(mid, mid, 3, 1, d, d)]
    main += [_bn(mid), nn.PReLU(mid), nn.Conv2d(mid, cout, 1, 1, 0, bias=False), _bn(cout), ScaledDropout2d(p_drop)]
    skip = [Same()]
    if kind == "down":
        skip += [nn.MaxPool2d(2, 2, 0), ChannelPad(cout - cin)]
    return nn.Sequential(Branches(nn.Sequential(*main), nn.Sequential(*skip)), Join("add"), nn.PReLU(cout))


_STAGE23 = (("plain", 0), ("dilated", 2), ("asym", 0), ("dilated", 4), ("plain", 0), ("dilated", 8), ("asym", 0),
            ("dilated", 16))


def create_enet(num_classes):
    """lib/enet.py:130-695: initial block (3x3/2 conv to 13 channels next to the max-pooled image), stage 1 (down + 4
    plain bottlenecks, 64 channels, p = 0.01), stages 2 and 3 (down / 8 + 8 mixed bottlenecks, 128 channels, p = 0.1), 1x1
    classifier.  No decoder: the 3D pipeline reads the encoder."""
    layers = [Branches(nn.Conv2d(3, 13, 3, 2, 1), nn.MaxPool2d(2, 2, 0)), Join("cat"), _bn(16), nn.PReLU(16),
              _bottleneck(16, 64, "down", 0, 0.01)]
    layers += [_bottleneck(64, 64, "plain", 0, 0.01) for _ in range(4)]
    layers += [_bottleneck(64, 128, "down", 0, 0.1)]
    layers += [_bottleneck(128, 128, k, a, 0.1) for k, a in _STAGE23 + _STAGE23]
    layers += [nn.Sequential(nn.Conv2d(128, num_classes, 1, 1, 0, bias=False))]
    return nn.Sequential(*layers)


def create_enet_for_3d(num_2d_classes, model_path, num_3d_classes=None):
    """lib/enet.py:697-717: (frozen front, the last 8 bottlenecks, classifier) of the 2D-pretrained network.  model_path
    None: random weights (tests)."""
    model = create_enet(num_2d_classes)
    if model_path is not None:
        model.load_state_dict(torch.load(model_path, map_location="cpu"))
    n = len(model)
    fixed = nn.Sequential(*(model[i] for i in range(n - 9)))
    trainable = nn.Sequential(*(model[i] for i in range(n - 9, n - 1)))
    classifier = nn.Sequential(model[n - 1])
    for p in fixed.parameters():
        p.requires_grad = False
    return fixed, trainable, classifier


def feature_extractor(model_path, num_2d_classes=41, device=None):
    """the network compute_multiview_features.py:88-98 runs: front + trainable part, eval, no gradients"""
    fixed, trainable, _ = create_enet_for_3d(num_2d_classes, model_path)
    net = nn.Sequential(fixed, trainable).eval()
    for p in net.parameters():
        p.requires_grad = False
    return net.to(device) if device is not None else net


def preprocess_frames(frames, image_dims=IMAGE_DIMS):
    """frames (N, H, W, 3) uint8 on any device -> (N, 3, image_dims[1], image_dims[0]) fp32: nearest-neighbour resize to
    the target HEIGHT keeping the aspect ratio, centre crop to the target width, / 255, normalise
    (compute_multiview_features.py:58-78; PIL's nearest rule: source index = floor((i + 0.5) * in / out))."""
    N, H, W, _ = frames.shape
    tw, th = image_dims
    if (W, H) != (tw, th):
        rw = int(math.floor(th * float(W) / float(H)))
        dev = frames.device
        ys = ((torch.arange(th, device=dev, dtype=torch.float64) + 0.5) * (H / th)).floor().long().clamp_(max=H - 1)
        xs = ((torch.arange(rw, device=dev, dtype=torch.float64) + 0.5) * (W / rw)).floor().long().clamp_(max=W - 1)
        left = int(round((rw - tw) / 2.0))
        if left < 0:
            raise ValueError("preprocess_frames: the resized frame (%d wide) is narrower than the crop (%d)" % (rw, tw))
        frames = frames[:, ys][:, :, xs[left:left + tw]]
    x = frames.permute(0, 3, 1, 2).float() / 255.0
    mean = torch.tensor(MEAN, device=x.device).view(1, 3, 1, 1)
    std = torch.tensor(STD, device=x.device).view(1, 3, 1, 1)
    return (x - mean) / std


@torch.no_grad()
def extract_features(net, frames, batch_size=256):
    """frames (N, H, W, 3) uint8 -> (N, 128, 32, 41) fp32 feature maps, `batch_size` frames per pass (:104,108-109)"""
    dev = next(net.parameters()).device
    out = []
    for i in range(0, frames.shape[0], batch_size):
        out.append(net(preprocess_frames(frames[i:i + batch_size].to(dev))))
    return torch.cat(out, 0)

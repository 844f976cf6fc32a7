"""SharedMLP (1x1 conv -> BatchNorm -> ReLU stacks) and the BN-momentum scheduler -- mirror of
the live part of the reference's lib/pointnet2/pytorch_utils.py:11-36,76-225,294-335.

State-dict layout is part of the drop-in contract (SURVEY.md §5 "Checkpoint / resume"):
`layer{i}.conv.weight` (Cout,Cin,1,1), `layer{i}.bn.bn.{weight,bias,running_mean,running_var,
num_batches_tracked}`.  Conv bias is dropped when bn=True (pytorch_utils.py:124), conv weights are
kaiming-normal, BN weight 1 / bias 0 (:78-83,133-135).
"""
import torch
import torch.nn as nn


class _BN(nn.Sequential):
    """`bn.bn` nesting of the reference (_BNBase, pytorch_utils.py:76-83)."""

    def __init__(self, num_features, kind):
        super().__init__()
        self.add_module("bn", kind(num_features))
        nn.init.constant_(self[0].weight, 1.0)
        nn.init.constant_(self[0].bias, 0.0)


class BatchNorm1d(_BN):
    def __init__(self, in_size, *, name=""):
        super().__init__(in_size, nn.BatchNorm1d)


class BatchNorm2d(_BN):
    def __init__(self, in_size, name=""):
        super().__init__(in_size, nn.BatchNorm2d)


class Conv2d(nn.Sequential):
    """conv(1x1) [-> bn] [-> activation]; children named conv / bn / activation (:104-157, :195-225)."""

    def __init__(self, in_size, out_size, *, kernel_size=(1, 1), stride=(1, 1), padding=(0, 0),
                 activation=nn.ReLU(inplace=True), bn=False, init=nn.init.kaiming_normal_, bias=True,
                 preact=False, name=""):
        super().__init__()
        if preact:
            raise NotImplementedError("preact SharedMLP has no caller in BridgeQA")
        bias = bias and (not bn)
        conv = nn.Conv2d(in_size, out_size, kernel_size=kernel_size, stride=stride, padding=padding, bias=bias)
        init(conv.weight)
        if bias:
            nn.init.constant_(conv.bias, 0)
        self.add_module(name + "conv", conv)
        if bn:
            self.add_module(name + "bn", BatchNorm2d(out_size))
        if activation is not None:
            self.add_module(name + "activation", activation)


class _BNReLUPointMajor(torch.autograd.Function):
    """Training-mode BatchNorm2d + ReLU (+ max over nsample) on the NHWC convolution output, csrc/bn.hip.
    x: (B, C, M, S) bf16 in channels_last memory = rows (b, m, s) x C.  Returns (B, C, M, S) channels_last bf16, or
    with pool the point-major (B, M, C) bf16 maxima (pointnet2_modules.py:259-262)."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, num_batches_tracked, eps, momentum, relu, pool):
        from . import _ext
        B, C, M, S = x.shape
        rows = x.permute(0, 2, 3, 1).reshape(B * M * S, C)  # a view: channels_last IS row-major (b, m, s, c)
        y, stats = _ext.bn_relu_fwd(rows, weight, bias, running_mean, running_var, num_batches_tracked, eps, momentum,
                                    S, relu, pool)
        ctx.save_for_backward(rows, stats)
        ctx.cfg = (B, C, M, S, relu, pool)
        if pool:
            return y.view(B, M, C)
        return y.view(B, M, S, C).permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dy):
        from . import _ext
        rows, stats = ctx.saved_tensors
        B, C, M, S, relu, pool = ctx.cfg
        if pool:
            dy2 = dy.contiguous().view(B * M, C)
        else:
            dy2 = dy.permute(0, 2, 3, 1)
            dy2 = (dy2 if dy2.is_contiguous() else dy2.contiguous()).view(B * M * S, C)
        if dy2.dtype != torch.bfloat16:
            dy2 = dy2.to(torch.bfloat16)
        dx, dgamma, dbeta = _ext.bn_relu_bwd(dy2, rows, stats, S, relu, pool)
        return dx.view(B, M, S, C).permute(0, 3, 1, 2), dgamma, dbeta, None, None, None, None, None, None, None


def _bn_kernel_ok(x, layer):
    """the fused BN+ReLU kernels cover: training-mode BatchNorm2d with a momentum, ReLU (or no) activation, bf16 NHWC
    activations with a power-of-two channel count"""
    if not hasattr(layer, "bn"):
        return False
    bn = layer.bn.bn
    act = getattr(layer, "activation", None)
    C = bn.num_features
    return (bn.training and bn.momentum is not None and bn.track_running_stats and bn.affine
            and (act is None or isinstance(act, nn.ReLU)) and C >= 8 and (C & (C - 1)) == 0 and C <= 2048
            and bn.weight.dtype == torch.float32 and x.is_cuda)


class SharedMLP(nn.Sequential):
    """args=[C0,C1,...,Ck] -> k layers named layer0..layer{k-1} (pytorch_utils.py:11-36).
"""

    def __init__(self, args, *, bn=False, activation=nn.ReLU(inplace=True), preact=False, first=False, name=""):
        super().__init__()
        for i in range(len(args) - 1):
            self.add_module(name + "layer{}".format(i),
                            Conv2d(args[i], args[i + 1], bn=bn, activation=activation, preact=preact))

    def forward(self, x, pool=False):
        """pool=True (point-major bf16 path only): also take the max over the last axis (nsample) and return the
        point-major (B, M, C) result -- fused into the last layer's BatchNorm+ReLU kernel."""
        if x.dtype != torch.bfloat16:
            assert not pool
            return super().forward(x)  # reference composition (fp32: conv1x1 -> BN -> ReLU per layer)
        if x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last):
            # bf16 NHWC grouped tensor (point-major fast path): the 1x1 convolutions run as bf16 implicit GEMMs on the
            # layout as it is; BatchNorm (fp32 parameters and statistics) + ReLU (+ the max over nsample) are one
            # stats pass and one apply pass over the convolution output (csrc/bn.hip)
            from . import pointnet2_utils
            last = len(self) - 1
            for i, layer in enumerate(self):
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    y = layer.conv(x)
                if (pointnet2_utils.backend_is_hip() and _bn_kernel_ok(y, layer)
                        and y.is_contiguous(memory_format=torch.channels_last) and y.dtype == torch.bfloat16):
                    bn = layer.bn.bn
                    x = _BNReLUPointMajor.apply(y, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                                bn.num_batches_tracked, bn.eps, bn.momentum,
                                                hasattr(layer, "activation"), pool and i == last)
                else:
                    with torch.autocast("cuda", dtype=torch.bfloat16):
                        for name, mod in layer.named_children():
                            if name != "conv":
                                y = mod(y)
                    x = y
                    if pool and i == last:
                        x = x.permute(0, 2, 3, 1).max(dim=2)[0]
            return x
        assert not pool
        # bf16 path: each 1x1 convolution is the batched GEMM  W[Cout,Cin] @ X[b][Cin, positions]  on the
        # channel-major layout (bf16 operands, fp32 accumulation; no MIOpen NCHW<->NHWC transposes);
        # BatchNorm statistics in fp32.
        from . import fusion_ops
        shape = x.shape
        x = x.flatten(2)
        for layer in self:
            conv = layer.conv
            x = torch.matmul(fusion_ops._c(conv.weight).flatten(1), x)
            if conv.bias is not None:
                x = x + fusion_ops._c(conv.bias)[None, :, None]
            if hasattr(layer, "bn"):
                bn = layer.bn.bn
                if bn.training and bn.num_batches_tracked is not None:
                    bn.num_batches_tracked.add_(1)
                x = nn.functional.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.training,
                                             bn.momentum if bn.momentum is not None else 0.0, bn.eps)
            if hasattr(layer, "activation"):
                x = nn.functional.relu(x, inplace=True)
        return x.view(shape[0], x.shape[1], *shape[2:])


def set_bn_momentum_default(bn_momentum):
    def fn(m):
        if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)):
            m.momentum = bn_momentum
    return fn


class BNMomentumScheduler(object):
    """Used by the reference's solver (lib/solver.py:24,271-277); pytorch_utils.py:303-335."""

    def __init__(self, model, bn_lambda, last_epoch=-1, setter=set_bn_momentum_default):
        if not isinstance(model, nn.Module):
            raise RuntimeError("Class '{}' is not a PyTorch nn Module".format(type(model).__name__))
        self.model, self.setter, self.lmbd = model, setter, bn_lambda
        self.step(last_epoch + 1)
        self.last_epoch = last_epoch

    def step(self, epoch=None):
        if epoch is None:
            epoch = self.last_epoch + 1
        self.last_epoch = epoch
        self.model.apply(self.setter(self.lmbd(epoch)))

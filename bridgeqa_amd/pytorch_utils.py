"""SharedMLP (1x1 conv -> BatchNorm -> ReLU stacks) and the BN-momentum scheduler -- mirror of
the live part of the reference's lib/pointnet2/pytorch_utils.py:11-36,76-225,294-335.

State-dict layout is part of the drop-in contract (SURVEY.md §5 "Checkpoint / resume"):
`layer{i}.conv.weight` (Cout,Cin,1,1), `layer{i}.bn.bn.{weight,bias,running_mean,running_var,
num_batches_tracked}`.  Conv bias is dropped when bn=True (pytorch_utils.py:124), conv weights are
kaiming-normal, BN weight 1 / bias 0 (:78-83,133-135).
"""
import os

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


class _ConvBNReLUPointMajor(torch.autograd.Function):
    """One SharedMLP layer -- 1x1 convolution (bias=False) -> training-mode BatchNorm2d -> ReLU (-> max over nsample) --
    on point-major bf16 rows, all on this repo's kernels (reference lib/pointnet2/pytorch_utils.py:104-157, 11-36;
    pointnet2_modules.py:259-262 for the pooling):
      forward : csrc/gemm.hip pwconv64_kernel  y = x W^T with the BatchNorm statistics taken from its fp32 accumulators
                (no statistics pass over y, no library convolution), then csrc/bn.hip bn_apply (normalise, clamp, pool);
      backward: csrc/bn.hip bn_backward (dgamma, dbeta, gradient w.r.t. y), then the MFMA GEMM family of csrc/gemm.hip
                for dX = dY W (transposed LDS reads of W) and dW = dY^T X (the contraction over the millions of rows cut
                into pieces that accumulate with fp32 atomics).
    x: (R, ldx) rows view (contiguous elements, ldx % 8 == 0, the first K elements are the input channels)."""

    @staticmethod
    def forward(ctx, x, conv_weight, gamma, beta, running_mean, running_var, num_batches_tracked, eps, momentum, relu,
                pool, S, conv_bias=None, x_stats=None, defer=False, link_in=None, link_out=None):
        """Deferred activations (round 5, SharedMLP.forward wires them): with `defer` the layer skips its BatchNorm + ReLU
        pass and returns (its stored pre-activation, its stats) -- the NEXT layer, called with x = that pre-activation and
        x_stats = those stats, applies relu(x scale + shift) tile by tile inside its convolution kernel and, in the backward,
        inside the fused pass (csrc/gemm.hip pwconv64s_kernel<.., true>, csrc/detbwd.hip sa_bwd_kernel<.., XT>).  By
        convention the gradient such a layer receives for its first output is the gradient w.r.t. the ACTIVATION (what the
        next layer's fused backward computes as its dX), exactly what this backward expects as dout."""
        from . import _ext, fusion_ops
        K = conv_weight.shape[1]
        w_pad = fusion_ops.padded_conv_shadow(conv_weight)
        # (the pre-activation is stored as its deviation from running_mean -- last steps' estimate of the channel mean:
        # BatchNorm's output does not depend on the offset, the bf16 rounding of the stored tensor does; CENTER_PREACT)
        # (a pooled layer whose backward will run fused records the arg-max of every group: csrc/detbwd.hip reads the table
        # instead of searching the group again)
        want_arg = bool(pool and _ext.FUSED_SA_BWD[0] and x.shape[0] >= _WGRAD_ROWS_MIN
                        and _ext.sa_bwd_supported(x.stride(0), conv_weight.shape[0], S, True, ctx.needs_input_grad[0]))
        out, y_raw, stats, arg = _ext.pwconv_bn_relu_fwd(x, K, w_pad, gamma, beta, running_mean, running_var,
                                                         num_batches_tracked, eps, momentum, S, relu, pool,
                                                         center=running_mean if CENTER_PREACT[0] else None,
                                                         want_arg=True, x_stats=x_stats, defer_apply=defer) if want_arg else (
            _ext.pwconv_bn_relu_fwd(x, K, w_pad, gamma, beta, running_mean, running_var, num_batches_tracked, eps,
                                    momentum, S, relu, pool, center=running_mean if CENTER_PREACT[0] else None,
                                    x_stats=x_stats, defer_apply=defer) + (None,))
        if conv_bias is not None and running_mean is not None:
            # a convolution bias in front of a training-mode BatchNorm cancels in the normalised output (it shifts the
            # batch mean by itself) and its gradient is identically zero; the only trace it leaves is in running_mean
            running_mean.add_(conv_bias.detach(), alpha=float(momentum))
        ctx.save_for_backward(x, w_pad, y_raw, stats, arg, x_stats)
        ctx.cfg = (K, S, relu, pool)
        ctx.conv_weight, ctx.conv_bias = conv_weight, conv_bias
        # link_in / link_out: dicts shared with the previous / next layer of the SharedMLP -- the next layer's fused backward
        # leaves this layer's dbeta | dgamma in link_out["dgb"] (the reduction rides on its pass: bq_sa_bwd_fused_xr)
        ctx.link_in, ctx.link_out = link_in, link_out
        if defer:
            ctx.mark_non_differentiable(stats)
            return y_raw, stats
        return out

    @staticmethod
    def backward(ctx, dout, _dstats=None):
        from . import _ext
        x, w_pad, y_raw, stats, arg, x_stats = ctx.saved_tensors
        K, S, relu, pool = ctx.cfg
        R, ldx = x.shape[0], x.stride(0)
        N = y_raw.shape[1]
        dout = dout.contiguous() if not dout.is_contiguous() else dout
        if dout.dtype != torch.bfloat16:
            dout = dout.to(torch.bfloat16)
        need_dx, need_dw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if (_ext.FUSED_SA_BWD[0] and need_dw and R >= _WGRAD_ROWS_MIN and (not pool or arg is not None)
                and (x_stats is None or need_dx)
                and _ext.sa_bwd_supported(ldx, N, S, pool, need_dx) and R * max(ldx, N) * 2 < (1 << 31) - (1 << 20)):
            # the BatchNorm reduction, then ONE pass over the activations for dX and dW (csrc/detbwd.hip)
            dgb = ctx.link_out.pop("dgb", None) if ctx.link_out is not None else None   # (summed by the next layer's pass)
            if dgb is None:
                dgb = _ext.bn_bwd_reduce(dout, y_raw, stats, S, relu, pool, arg)
            xs = torch.as_strided(x, (R, ldx), (ldx, 1))
            carry = x_stats is not None and ctx.link_in is not None and _ext.CARRY_REDUCE[0]
            res = _ext.sa_bwd_fused(xs, y_raw, dout, arg, w_pad, stats, dgb, S, relu, pool, need_dx, x_stats=x_stats,
                                    carry_reduce=carry)
            dx_full, dwf = res[0], res[1]
            if carry and res[2] is not None:
                ctx.link_in["dgb"] = res[2]
            dx = dx_full[:, :x.shape[1]] if need_dx else None
            dw = dwf[:, :K].reshape(ctx.conv_weight.shape)
            dcb = torch.zeros_like(ctx.conv_bias) if ctx.conv_bias is not None else None
            return dx, dw, dgb[1], dgb[0], None, None, None, None, None, None, None, None, dcb, None, None, None, None
        if x_stats is not None:
            # (a deferred input on the four-kernel path -- the fused kernel switched off between forward and backward, or an
            # operand beyond its 2 GB bound: the activation is formed once, now)
            x = _ext.bn_apply(x, x_stats, 1, True, False)
        dy, dgamma, dbeta = _ext.bn_relu_bwd(dout, y_raw, stats, S, relu, pool)
        dx = None
        if ctx.needs_input_grad[0]:
            # dX (R, ldx) = dY (R, N) W (N, ldx): the padding columns of W are zero, so are the padding columns of dX
            dx_full = torch.empty(R, ldx, dtype=torch.bfloat16, device=x.device)
            _ext.gemm_grouped([dict(P=w_pad[:, :ldx], Q=dy, out=dx_full)], _ext.GEMM_P_XC, _ext.EPI_NONE, 64)
            dx = dx_full[:, :x.shape[1]]
        dw = None
        if ctx.needs_input_grad[1]:
            xs = torch.as_strided(x, (R, ldx), (ldx, 1))  # whole padded rows (the padding is zero / zero-weighted)
            tiles = ((ldx + 63) // 64) * (N // 64)
            if _WGRAD_ROWS and R >= _WGRAD_ROWS_MIN and _ext.wgrad_rows_ok(ldx, N):
                # whole rows staged once for all output tiles, slices summed in a fixed order: no atomics
                dwf = _ext.wgrad_rows(xs, dy, torch.empty(N, ldx, dtype=torch.float32, device=x.device), _WGRAD_ROWS_WGS)
            elif _CUT_DW_DETERMINISTIC[0]:
                dwf, _ = _cut_dw(xs, dy, _wgrad_pieces(R, tiles), False)
            else:
                dwf = torch.zeros(N, ldx, dtype=torch.float32, device=x.device)
                ksplit = _wgrad_pieces(R, tiles)
                _ext.gemm_grouped([dict(P=xs, Q=dy, out=dwf, ksplit=ksplit)],
                                  _ext.GEMM_P_XC | _ext.GEMM_Q_XC | _ext.GEMM_OUT_F32, _ext.EPI_NONE, 64)
            dw = dwf[:, :K].reshape(ctx.conv_weight.shape)
        dcb = torch.zeros_like(ctx.conv_bias) if ctx.conv_bias is not None else None
        if ctx.link_out is not None:
            ctx.link_out.pop("dgb", None)
        return dx, dw, dgamma, dbeta, None, None, None, None, None, None, None, None, dcb, None, None, None, None


CENTER_PREACT = [True]      # SharedMLP pre-activations stored relative to running_mean (tools/loss_gap_probe.py)
_WGRAD_ROWS = True          # the whole-row weight-gradient kernel for long contractions
_WGRAD_ROWS_MIN = 65536
_WGRAD_ROWS_WGS = 0     # 0: the library's default per shape
_WGRAD_WGS = 384   # (sweep in the c3 step: 398 / 405 / 406 / 406 / 404 / 388 samples/s at 128 / 256 / 384 / 512 / 1536 / 3072)


def _wgrad_pieces(R, tiles):
    """pieces the row contraction of a layer's weight gradient is cut into: ~_WGRAD_WGS workgroups in all (the 64-tile
    kernel keeps 3 per CU resident: 768 a round), a multiple of 8 so that the kernel's XCD-aware order applies"""
    ks = max(1, min((R + 63) // 64, _WGRAD_WGS // tiles))
    return ks - ks % 8 if ks >= 8 else ks


_CUT_DW_DETERMINISTIC = [True]   # cut weight-gradient contractions as separate problems summed in a fixed order (no fp32 atomics)


def _cut_dw(xs, dy, pieces, want_colsum):
    """dW (N, ld) = dy^T xs -- and, on request, the column sums of dy -- with the row contraction cut into `pieces` PROBLEMS of one
    grouped launch (whole K tiles each, at most the launch's 36), whose fp32 partial results are summed in a fixed order: no
    atomics, bit-reproducible (round 6: these cut contractions were the last gradients of the detector that moved from one
    execution to the next, tools/grad_determinism.py).  xs (R, ld), dy (R, N) bf16 rows."""
    from . import _ext
    R, N, ld = xs.shape[0], dy.shape[1], xs.shape[1]
    pieces = max(1, min(int(pieces), int(_ext._lib.bq_gemm_max_problems())))
    step = -(-(-(-R // pieces)) // 64) * 64
    pieces = -(-R // step)
    part = torch.empty(pieces, N, ld, dtype=torch.float32, device=xs.device)
    cs = torch.empty(pieces, N, dtype=torch.float32, device=xs.device) if want_colsum else None
    probs = []
    for i in range(pieces):
        r0, r1 = i * step, min(R, (i + 1) * step)
        pr = dict(P=xs[r0:r1], Q=dy[r0:r1], out=part[i])
        if want_colsum:
            pr["colsum"] = cs[i]
        probs.append(pr)
    _ext.gemm_grouped(probs, _ext.GEMM_P_XC | _ext.GEMM_Q_XC | _ext.GEMM_OUT_F32, _ext.EPI_NONE, 64)
    if pieces == 1:
        return part[0], (cs[0] if want_colsum else None)
    return part.sum(0), (cs.sum(0) if want_colsum else None)


def _rows_view_ok(x):
    """the uniform row stride of a (B, C, M, S) logical NCHW tensor that is physically point-major rows (b, m, s) with
    contiguous channels (ld >= C, a multiple of 8 elements, 16-byte aligned base), or None"""
    if x.dim() != 4 or x.stride(1) != 1:
        return None
    B, C, M, S = x.shape
    ld = x.stride(3)
    if ld < C or ld % 8 or x.stride(2) != S * ld or x.stride(0) != M * S * ld or x.data_ptr() % 16:
        return None
    return ld


def _rows_view(x):
    """-> the (B*M*S, C) rows view with stride (ld, 1) of such a tensor, or None"""
    ld = _rows_view_ok(x)
    if ld is None:
        return None
    B, C, M, S = x.shape
    if x.requires_grad and torch.is_grad_enabled():
        return _RowsViewFn.apply(x, ld)
    return torch.as_strided(x, (B * M * S, C), (ld, 1))


class _RowsViewFn(torch.autograd.Function):
    """the (B*M*S, C) rows view of _rows_view with a backward that is a VIEW too: the rows' gradient arrives as the first C
    columns of the padded (R, ld') buffer the first layer's dX GEMM wrote and goes on as (B, C, M, S) strides over that same
    buffer (pointnet2_utils._GroupConcatPM's gradient kernel reads padded rows).  torch.as_strided's own backward
    zero-fills a tensor of the base's size and copies the gradient into it: 142 + 35 + 17 MB of fills and three full copies
    per c3 step on the detector stream."""

    @staticmethod
    def forward(ctx, x, ld):
        B, C, M, S = x.shape
        ctx.dims = (B, C, M, S)
        return torch.as_strided(x, (B * M * S, C), (ld, 1))

    @staticmethod
    def backward(ctx, g):
        B, C, M, S = ctx.dims
        if g.stride(1) != 1 or g.stride(0) < C:
            g = g.contiguous()
        lg = g.stride(0)
        return torch.as_strided(g, (B, C, M, S), (M * S * lg, 1, S * lg, lg)), None


def rows_conv_bn_relu(rows, conv, bn, relu=True):
    """1x1 Conv1d / Conv2d (+ bias) -> training-mode BatchNorm -> ReLU on point-major bf16 rows (R, K) through the native
    layer; None when its preconditions do not hold (the caller then runs the reference composition).  Used by the
    feature-propagation SharedMLPs, the voting module and the proposal head (reference lib/pointnet2/
    pointnet2_modules.py:330-340, models/voting_module.py:33-40, models/proposal_module.py:44-50)."""
    if not (rows.is_cuda and rows.dtype == torch.bfloat16 and rows.dim() == 2 and rows.stride(1) == 1
            and rows.stride(0) % 8 == 0 and rows_layer_ok(conv, bn)):
        return None
    return _ConvBNReLUPointMajor.apply(rows, conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                       bn.num_batches_tracked, bn.eps, bn.momentum, relu, False, 1, conv.bias)


class _RowsLinearF32(torch.autograd.Function):
    """y (R, N) fp32 = rows (R, K) bf16 @ W^T + b for an output layer whose width is not a kernel-friendly number (the
    voting module's 3 + 256 = 259 channels, voting_module.py:27-31; the proposal head's 2 + 3 + 2 NH + 4 NS + NC = 97,
    proposal_module.py:48-56) through csrc/gemm.hip: the weight is zero-padded to a multiple of 8 output rows (64 for the
    contraction of the input gradient), the result keeps fp32 (these are the regression outputs the losses read).
    Backward: dX = dY W (bf16), dW = dY^T X and db = column sums of dY from ONE launch (cut contraction, fp32 atomics)."""

    @staticmethod
    def forward(ctx, rows, weight, bias):
        from . import _ext
        N, K = weight.shape
        Np, Nc = (N + 7) // 8 * 8, (N + 63) // 64 * 64
        wp = torch.zeros(Nc, K, dtype=torch.bfloat16, device=rows.device)
        wp[:N].copy_(weight)
        bp = None
        if bias is not None:
            bp = torch.zeros(Np, dtype=torch.float32, device=rows.device)
            bp[:N].copy_(bias)
        out = torch.empty(rows.shape[0], Np, dtype=torch.float32, device=rows.device)
        _ext.gemm_grouped([dict(P=wp[:Np], Q=rows, out=out, bias=bp)], _ext.GEMM_OUT_F32,
                          _ext.EPI_BIAS if bp is not None else _ext.EPI_NONE, 64)
        ctx.save_for_backward(rows, wp)
        ctx.dims = (N, K, Np, Nc, bias is not None)
        return out[:, :N]

    @staticmethod
    def backward(ctx, g):
        from . import _ext
        rows, wp = ctx.saved_tensors
        N, K, Np, Nc, has_bias = ctx.dims
        R = rows.shape[0]
        gp = torch.zeros(R, Np, dtype=torch.bfloat16, device=g.device)
        gp[:, :N].copy_(g)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            # contraction over the Nc zero-padded weight rows: a row of gp (Np elements) runs into the next one, finite
            # values against zero weights (include/bqhip_fusion.h, p_bytes / q_bytes)
            dx = torch.empty(R, K, dtype=torch.bfloat16, device=g.device)
            _ext.gemm_grouped([dict(P=wp, Q=gp, out=dx, Kc=Nc, q_bytes=gp.numel() * 2)], _ext.GEMM_P_XC, _ext.EPI_NONE)
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            if _CUT_DW_DETERMINISTIC[0]:
                dwp, dbp = _cut_dw(rows, gp, max(1, min(64, R // 512)), True)
            else:
                dwp = torch.zeros(Np, K, dtype=torch.float32, device=g.device)
                dbp = torch.zeros(Np, dtype=torch.float32, device=g.device)
                _ext.gemm_grouped([dict(P=rows, Q=gp, out=dwp, colsum=dbp, ksplit=max(1, min(64, R // 512)))],
                                  _ext.GEMM_P_XC | _ext.GEMM_Q_XC | _ext.GEMM_OUT_F32, _ext.EPI_NONE, 64)
            dw, db = dwp[:N], (dbp[:N] if has_bias else None)
        return dx, dw, db


def rows_linear_f32(rows, weight, bias):
    """fp32 (R, N) = bf16 rows (R, K) @ weight (N, K)^T + bias through the MFMA GEMM family (any N); see _RowsLinearF32"""
    return _RowsLinearF32.apply(rows, weight, bias)


def rows_layer_ok(conv, bn):
    """the layer-side preconditions of rows_conv_bn_relu: a caller that chains several layers checks ALL of them before
    running the first (a fallback after a native layer has run would update that layer's running statistics twice)"""
    C = bn.num_features
    return (bn.training and bn.momentum is not None and bn.track_running_stats and bn.affine
            and C % 64 == 0 and (C & (C - 1)) == 0 and conv.weight.dtype == torch.float32
            and all(k == 1 for k in conv.kernel_size))


def to_rows(x):
    """(B, C, n) channel-major features -> (B*n, C) bf16 point-major rows (one transposing cast)"""
    B, C, n = x.shape
    return x.transpose(1, 2).to(torch.bfloat16).contiguous().view(B * n, C)


def native_rows_ok(x):
    """the row-form native layers apply: bf16 compute dtype on a GPU with the HIP backend"""
    from . import fusion_ops, pointnet2_utils
    return (x.is_cuda and fusion_ops.compute_dtype() == torch.bfloat16 and fusion_ops.POINT_MAJOR[0]
            and pointnet2_utils.backend_is_hip())


def _native_layer_ok(layer):
    """the native SharedMLP layer covers: conv without bias, training-mode BatchNorm2d with momentum and affine
    parameters, ReLU or no activation, output channels a multiple of 64 and a power of two (the BN kernels)"""
    if not hasattr(layer, "bn") or layer.conv.bias is not None:
        return False
    bn = layer.bn.bn
    act = getattr(layer, "activation", None)
    C = bn.num_features
    return (bn.training and bn.momentum is not None and bn.track_running_stats and bn.affine
            and (act is None or isinstance(act, nn.ReLU)) and C % 64 == 0 and (C & (C - 1)) == 0 and C <= 2048
            and bn.weight.dtype == torch.float32 and layer.conv.weight.dtype == torch.float32
            and tuple(layer.conv.kernel_size) == (1, 1))


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


def _defer_ok(rows, layer, nxt, S, nxt_pool):
    """may `layer` hand its pre-activation to `nxt` instead of writing its activation?  Both native, ReLU, 64 or 128 channels
    in between, enough rows for the fused backward and a fused kernel for the next layer's shape"""
    from . import _ext
    N, R = layer.conv.weight.shape[0], rows.shape[0]
    return bool(_ext.DEFER_BN[0] and _ext.FUSED_SA_BWD[0] and not _ext.FP32_PREACT[0] and torch.is_grad_enabled()
                and hasattr(layer, "activation") and N in (64, 128) and nxt.conv.weight.shape[1] == N
                and R >= _WGRAD_ROWS_MIN and R * max(N, nxt.conv.weight.shape[0]) * 2 < (1 << 31) - (1 << 20)
                and (not nxt_pool or int(S) in (16, 32, 64))
                and _ext.sa_bwd_supported(N, nxt.conv.weight.shape[0], S, nxt_pool, True))


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
        from . import pointnet2_utils
        rows = _rows_view(x) if (x.is_cuda and pointnet2_utils.backend_is_hip()) else None
        if rows is not None and all(_native_layer_ok(layer) for layer in self):
            # point-major rows (csrc/pn2_ops.hip group_concat_pm) all the way: every layer is one GEMM launch with the
            # BatchNorm statistics in its epilogue + one normalise / ReLU (/ max over nsample) pass; no library
            # convolution, no layout change, no statistics pass (see _ConvBNReLUPointMajor)
            B, _, M, S = x.shape
            last = len(self) - 1
            layers = list(self)
            x_stats = link = None
            for i, layer in enumerate(layers):
                bn = layer.bn.bn
                # (round 5) the BatchNorm + ReLU between two convolutions is not materialised where the next layer can apply
                # it on load, forward and backward: this layer hands over its stored pre-activation and its stats
                defer = i < last and _defer_ok(rows, layer, layers[i + 1], S, pool and i + 1 == last)
                nxt_link = {} if defer else None
                res = _ConvBNReLUPointMajor.apply(rows, layer.conv.weight, bn.weight, bn.bias, bn.running_mean,
                                                  bn.running_var, bn.num_batches_tracked, bn.eps, bn.momentum,
                                                  hasattr(layer, "activation"), pool and i == last, S, None, x_stats, defer,
                                                  link, nxt_link)
                rows, x_stats = res if defer else (res, None)
                link = nxt_link
            C = rows.shape[1]
            return rows.view(B, M, C) if pool else rows.view(B, M, S, C).permute(0, 3, 1, 2)
        if rows is not None and not x.is_contiguous(memory_format=torch.channels_last):
            x = x.contiguous(memory_format=torch.channels_last)  # padded point rows -> compact NHWC for the library path
        if x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last):
            # bf16 NHWC grouped tensor without the native layer's preconditions (eval-mode BatchNorm, odd channel
            # counts): the 1x1 convolutions run as library bf16 implicit GEMMs; BatchNorm + ReLU (+ pool) on csrc/bn.hip
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

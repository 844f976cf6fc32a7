"""Module-level parity on the GPU: bridgeqa_amd's layers over the HIP backend against the golden
vectors produced by the reference's Python layers (fp32 path; tolerance covers rocBLAS/MIOpen
summation order and fp32 atomics in the backward)."""
import pytest
import torch

from test_modules_cpu import run_c1, run_sa_fp_golden

pytestmark = pytest.mark.gpu


def test_sa_fp_modules_vs_reference_golden_on_gpu(golden, dev):
    from bridgeqa_amd import _ext, pointnet2_utils as pu
    assert pu.backend() is _ext
    run_sa_fp_golden(golden("pn2_modules.npz"), dev, 1e-4, 1e-4)


def test_backbone_voting_proposal_c1_on_gpu(golden, dev):
    run_c1(golden("pn2_backbone_c1.npz"), dev, 1e-3, 1e-3)


def test_backbone_c2_shapes_and_backward(dev):
    """BASELINE config 2 (B=16, N=40000): shapes of every data_dict entry, finite loss, grads reach SA1."""
    from bridgeqa_amd.backbone_module import Pointnet2Backbone
    from bridgeqa_amd.voting_module import VotingModule
    from conftest import scene
    torch.manual_seed(0)
    bb, vote = Pointnet2Backbone(input_feature_dim=1).to(dev), VotingModule(1, 256).to(dev)
    dd = bb({"point_clouds": scene(16, 40000, 1, 42).to(dev)})
    assert dd["sa1_features"].shape == (16, 128, 2048) and dd["fp2_features"].shape == (16, 256, 1024)
    assert dd["fp2_inds"].shape == (16, 1024) and dd["fp2_inds"].dtype == torch.int32
    vx, vf = vote(dd["fp2_xyz"], dd["fp2_features"])
    (vx.square().mean() + vf.square().mean()).backward()
    g = bb.sa1.mlp_module.layer0.conv.weight.grad
    assert g is not None and torch.isfinite(g).all() and g.abs().sum() > 0


def test_sa_module_bf16_path_vs_reference_golden(golden, dev):
    """bf16 grouped tensor + bf16 SharedMLP GEMMs (fp32 accumulation, fp32 BN statistics) against the fp32 reference
    golden: rel-L2 <= 1e-2 on the outputs, indices exact."""
    import numpy as np
    from bridgeqa_amd import fusion_ops
    from bridgeqa_amd.pointnet2_modules import PointnetSAModuleVotes
    from golden_util import fill_params
    g = golden("pn2_modules.npz")
    sa = PointnetSAModuleVotes(npoint=64, radius=0.9, nsample=16, mlp=[5, 16, 16, 32], use_xyz=True, normalize_xyz=True)
    fill_params(sa, "sa.")
    sa = sa.to(dev).train()
    pc = torch.from_numpy(g["sa_pc"]).to(dev)
    xyz = pc[..., :3].contiguous()
    feat = pc[..., 3:].transpose(1, 2).contiguous().requires_grad_(True)
    prev = fusion_ops.set_compute_dtype(torch.bfloat16)
    fusion_ops.SHAREDMLP_BF16[0] = True
    try:
        nx, nf, ni = sa(xyz, feat)
        (nf * torch.from_numpy(g["sa_w"]).to(dev)).sum().backward()
    finally:
        fusion_ops.SHAREDMLP_BF16[0] = False
        fusion_ops.set_compute_dtype(prev)
    assert nf.dtype == torch.float32
    np.testing.assert_array_equal(ni.cpu().numpy(), g["sa_inds"])
    rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
    assert rel(nf.detach().cpu().numpy(), g["sa_train_new_features"]) < 1.5e-2
    assert rel(feat.grad.cpu().numpy(), g["sa_train_grad_features"]) < 0.15  # bf16 re-rounding moves max-pool winners


def test_point_major_bf16_detector_vs_fp32_reference_path(dev):
    """The whole DET slice through the point-major bf16 fast path (grouping of contiguous rows, NHWC bf16 SharedMLP)
    against the fp32 reference-layout path: identical sampling / neighbour indices, features within bf16 tolerance,
    gradients reach the first SA layer."""
    import bench
    from bridgeqa_amd import fusion_ops
    from bridgeqa_amd.hotpath import ScanQAHotPath
    torch.manual_seed(0)
    model = ScanQAHotPath(input_feature_dim=7, use_blip=False).to(dev).eval()
    pc = bench.synth_batch(2, 6000, 7, 3, dev)
    with torch.no_grad():
        ref = model.detect({"point_clouds": pc})
    prev = fusion_ops.set_compute_dtype(torch.bfloat16)
    try:
        with torch.no_grad():
            got = model.detect({"point_clouds": pc})
        model.train()
        labels = {k: v.to(dev) for k, v in bench.synth_labels(pc[..., :3].cpu(), 5).items()}
        dd = model.detect(dict({"point_clouds": pc}, **labels))
        bench.det_loss(dd).backward()
    finally:
        fusion_ops.set_compute_dtype(prev)
    for k in ("sa1_inds", "sa2_inds", "fp2_inds"):
        assert torch.equal(got[k], ref[k])
    rel = lambda a, b: ((a.float() - b.float()).norm() / b.float().norm()).item()
    # Tolerance: bf16 operands (8-bit mantissa) with fp32 accumulation.  With zero-mean random weights the dot
    # products cancel (|sum| ~ sqrt(K) of sum|.|), which amplifies the 0.2-0.4 % operand rounding to ~2 % per
    # SharedMLP; the error compounds over the 4 SA + 2 FP levels (measured 2.5 / 4.7 / ~6 %).
    for k, tol in (("sa1_features", 3e-2), ("sa2_features", 6e-2), ("sa4_features", 1e-1), ("fp2_features", 1e-1)):
        assert got[k].shape == ref[k].shape and got[k].is_contiguous()
        assert rel(got[k], ref[k]) < tol, (k, rel(got[k], ref[k]))
    g = model.detection_backbone.sa1.mlp_module.layer0.conv.weight.grad
    assert g is not None and torch.isfinite(g).all() and g.abs().sum() > 0


@pytest.mark.parametrize("B,C,M,S,relu,pool", [(2, 64, 37, 16, True, False), (2, 128, 50, 32, True, True),
                                               (1, 256, 9, 16, False, False), (3, 64, 700, 64, True, True),
                                               (2, 8, 5, 4, True, True)])
def test_fused_batchnorm_relu_maxpool_point_major(dev, B, C, M, S, relu, pool):
    """csrc/bn.hip vs the reference composition BatchNorm2d(train) -> ReLU -> max over nsample (pytorch_utils.py:104-157,
    pointnet2_modules.py:259-262) in fp32 on the same bf16-rounded input: outputs, running statistics,
    num_batches_tracked, and the gradients w.r.t. input, gamma, beta."""
    from bridgeqa_amd.pytorch_utils import _BNReLUPointMajor
    g = torch.Generator().manual_seed(B * 1000 + C + S)
    x = (torch.randn(B, C, M, S, generator=g) * 1.5 + 0.3).to(dev).to(torch.bfloat16)
    x = x.contiguous(memory_format=torch.channels_last)
    bn = torch.nn.BatchNorm2d(C, momentum=0.1).to(dev).train()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5); bn.bias.copy_(torch.randn(C, generator=g) * 0.2)
    ref_bn = torch.nn.BatchNorm2d(C, momentum=0.1).to(dev).train()
    ref_bn.load_state_dict(bn.state_dict())
    xa = x.clone().requires_grad_(True)
    y = _BNReLUPointMajor.apply(xa, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                bn.eps, bn.momentum, relu, pool)
    xr = x.float().requires_grad_(True)
    z = ref_bn(xr)
    if relu:
        z = torch.relu(z)
    if pool:
        z = z.max(dim=3)[0].transpose(1, 2)  # (B, M, C)
    assert y.shape == z.shape and y.dtype == torch.bfloat16
    dy = torch.randn(z.shape, generator=g).to(dev)
    y.backward(dy.to(torch.bfloat16))
    z.backward(dy.to(torch.bfloat16).float())
    rel = lambda a, b: ((a.float() - b.float()).norm() / (b.float().norm() + 1e-12)).item()
    assert rel(y, z) < 6e-3
    assert rel(bn.running_mean, ref_bn.running_mean) < 1e-4 and rel(bn.running_var, ref_bn.running_var) < 1e-4
    assert int(bn.num_batches_tracked) == 1
    assert rel(xa.grad, xr.grad) < 2e-2
    assert rel(bn.weight.grad, ref_bn.weight.grad) < 1e-2 and rel(bn.bias.grad, ref_bn.bias.grad) < 1e-2
    assert xa.grad.is_contiguous(memory_format=torch.channels_last)


def test_detection_losses_on_gpu_vs_reference_golden(golden, dev):
    """the reference's losses only run with .cuda() hard-wired; the mirror runs on the GPU without a host sync and
    matches the reference's own CPU-executed values (fp32 reductions: 1e-5)"""
    from test_loss_cpu import run_loss_golden
    run_loss_golden(golden, dev, 1e-5)


def test_fused_batchnorm_statistics_survive_a_large_mean(dev):
    """|mean| >> std over 2 M rows: a plain fp32 E[x^2] - mean^2 would lose the variance; the pivoted sums of
    csrc/bn.hip must not (compare with float64 statistics of the same bf16 values)."""
    from bridgeqa_amd import _ext
    R, C = 16 * 2048 * 64, 64
    g = torch.Generator().manual_seed(9)
    x = (torch.randn(R, C, generator=g) * 0.5 + torch.linspace(-40, 40, C)).to(dev).to(torch.bfloat16)
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    y, stats = _ext.bn_relu_fwd(x, gamma, beta, None, None, None, 1e-5, 0.1, 64, False, False)
    xd = x.double()
    mean, var = xd.mean(0), xd.var(0, unbiased=False)
    assert torch.allclose(stats[2].double(), mean, rtol=1e-5, atol=1e-4)
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    assert ((stats[3].double() - rstd).abs() / rstd).max().item() < 2e-3
    want = ((xd - mean) * rstd).float()
    assert ((y.float() - want).norm() / want.norm()).item() < 1e-2

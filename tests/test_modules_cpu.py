"""Host logic of bridgeqa_amd's Python layers, executed over the CPU oracle backend, against
golden vectors produced by the REFERENCE's Python layers (oracle/gen_golden.py).  This pins the
module API (ctor args, forward signatures, data_dict keys) and the state-dict key sets.  CPU only."""
import numpy as np
import pytest
import torch

from golden_util import fill_params, subsample


def keys_of(module, prefix):
    return ["%s %s" % (k, "x".join(map(str, s))) for k, s in fill_params(module, prefix)]


def close(a, g, rtol=1e-5, atol=1e-5):
    a = subsample(a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a)
    np.testing.assert_allclose(a, g, rtol=rtol, atol=atol)


def run_ops_golden(g, pu, to):
    for tag in ("n64", "n1000", "n4096"):
        xyz = to(g["fps_%s_xyz" % tag])
        inds = pu.furthest_point_sample(xyz, int(g["fps_%s_m" % tag]))
        np.testing.assert_array_equal(inds.cpu().numpy(), g["fps_%s_inds" % tag])
        new_xyz = pu.gather_operation(xyz.transpose(1, 2).contiguous(), inds).transpose(1, 2).contiguous()
        np.testing.assert_array_equal(new_xyz.cpu().numpy(), g["gather_%s_out" % tag])
        for key in [k for k in g if k.startswith("bq_%s_" % tag)]:
            r, S = key.split("_")[2:]
            got = pu.ball_query(float(r[1:]), int(S[1:]), xyz, new_xyz)
            np.testing.assert_array_equal(got.cpu().numpy(), g[key])
    feats = to(g["grp_feats"]).requires_grad_(True)
    out = pu.grouping_operation(feats, to(g["grp_idx"]))
    np.testing.assert_array_equal(out.detach().cpu().numpy(), g["grp_out"])
    out.backward(to(g["grp_go"]))
    close(feats.grad, g["grp_grad"])
    pts = to(g["gat_pts"]).requires_grad_(True)
    out = pu.gather_operation(pts, to(g["gat_inds"]))
    np.testing.assert_array_equal(out.detach().cpu().numpy(), g["gat_out"])
    out.backward(to(g["gat_go"]))
    close(pts.grad, g["gat_grad"], 1e-6, 1e-6)
    dist, idx = pu.three_nn(to(g["nn_unknown"]), to(g["nn_known"]))
    np.testing.assert_array_equal(idx.cpu().numpy(), g["nn_idx"])
    # the golden's sqrt is torch-CPU's vectorised sqrt, which is up to 1 ulp off; the HIP kernel's sqrt is
    # correctly rounded (checked exactly against numpy in test_ops_gpu.py) => 1-ulp tolerance here
    np.testing.assert_allclose(dist.cpu().numpy(), g["nn_dist"], rtol=1.3e-7, atol=0)
    kf = to(g["it_feats"]).requires_grad_(True)
    out = pu.three_interpolate(kf, idx, to(g["it_weight"]))
    np.testing.assert_array_equal(out.detach().cpu().numpy(), g["it_out"])
    out.backward(to(g["it_go"]))
    close(kf.grad, g["it_grad"])


def test_autograd_ops_vs_reference_golden(oracle_backend, golden):
    from bridgeqa_amd import pointnet2_utils as pu
    run_ops_golden(golden("pn2_ops.npz"), pu, torch.from_numpy)


def run_sa_fp_golden(g, dev, rtol, atol):
    from bridgeqa_amd.pointnet2_modules import PointnetFPModule, PointnetSAModuleVotes
    sa = PointnetSAModuleVotes(npoint=64, radius=0.9, nsample=16, mlp=[5, 16, 16, 32], use_xyz=True,
                               normalize_xyz=True)
    assert keys_of(sa, "sa.") == list(g["sa_keys"])
    sa = sa.to(dev)
    pc = torch.from_numpy(g["sa_pc"]).to(dev)
    xyz = pc[..., :3].contiguous()
    feat = pc[..., 3:].transpose(1, 2).contiguous().requires_grad_(True)
    sa.train()
    nx, nf, ni = sa(xyz, feat)
    np.testing.assert_array_equal(ni.cpu().numpy(), g["sa_inds"])
    np.testing.assert_array_equal(nx.cpu().numpy(), g["sa_train_new_xyz"])
    close(nf, g["sa_train_new_features"], rtol, atol)
    (nf * torch.from_numpy(g["sa_w"]).to(dev)).sum().backward()
    close(feat.grad, g["sa_train_grad_features"], rtol * 10, atol * 10)
    close(sa.mlp_module.layer0.conv.weight.grad, g["sa_train_grad_w0"], rtol * 10, atol * 10)
    close(sa.mlp_module.layer2.bn.bn.running_mean, g["sa_train_running_mean2"], rtol, atol)
    close(sa.mlp_module.layer2.bn.bn.running_var, g["sa_train_running_var2"], rtol, atol)
    sa.eval()
    _, nf, _ = sa(xyz, feat.detach())
    close(nf, g["sa_eval_new_features"], rtol, atol)
    fp = PointnetFPModule(mlp=[9 + 6, 16, 12])
    assert keys_of(fp, "fp.") == list(g["fp_keys"])
    fp = fp.to(dev).eval()
    t = lambda k: torch.from_numpy(g[k]).to(dev)
    out = fp(t("fp_unknown"), t("fp_known"), t("fp_unknown_feats"), t("fp_known_feats"))
    close(out, g["fp_out"], rtol, atol)


def test_sa_fp_modules_vs_reference_golden(oracle_backend, golden):
    run_sa_fp_golden(golden("pn2_modules.npz"), torch.device("cpu"), 1e-5, 1e-5)


def build_c1(g):
    from bridgeqa_amd.backbone_module import Pointnet2Backbone
    from bridgeqa_amd.proposal_module import ProposalModule
    from bridgeqa_amd.voting_module import VotingModule
    bb = Pointnet2Backbone(input_feature_dim=1)
    vote = VotingModule(1, 256)
    prop = ProposalModule(18, 1, 18, g["mean_size_arr"], 256, "vote_fps")
    keys = keys_of(bb, "detection_backbone.") + keys_of(vote, "voting_net.") + keys_of(prop, "proposal_net.")
    assert keys == list(g["keys"])  # state-dict key set and shapes == the reference's
    return bb, vote, prop


INT_KEYS = ("sa1_inds", "sa2_inds", "fp2_inds")  # FPS on the INPUT cloud / its prefix: exact
# vote-cluster FPS runs on network outputs (vote_xyz): a different fp32 summation order in the dense layers can
# flip a pick between near-tied votes, so those ids are compared as a set-valued property and the per-proposal
# tensors only on the proposals both sides picked at the same slot.
PROPOSAL_KEYS = ("aggregated_vote_xyz", "aggregated_vote_features", "objectness_scores", "center", "heading_scores",
                 "heading_residuals_normalized", "heading_residuals", "size_scores", "size_residuals_normalized",
                 "size_residuals", "sem_cls_scores", "bbox_corner", "bbox_feature", "bbox_mask", "bbox_sems")


def run_c1(g, dev, rtol, atol, modes=("eval", "train")):
    bb, vote, prop = [m.to(dev) for m in build_c1(g)]
    pc = torch.from_numpy(g["point_clouds"]).to(dev)
    for mode in modes:
        for m in (bb, vote, prop):
            m.train(mode == "train")
        dd = bb({"point_clouds": pc})
        vx, vf = vote(dd["fp2_xyz"], dd["fp2_features"])
        vf = vf.div(torch.norm(vf, p=2, dim=1).unsqueeze(1))  # qa_module.py:452-453
        dd["vote_xyz"], dd["vote_features"] = vx, vf
        dd = prop(vx, vf, dd)
        want = {k[len(mode) + 1:]: v for k, v in g.items() if k.startswith(mode + ".")}
        assert set(want) == set(dd) - {"point_clouds"}  # same data_dict keys as the reference writes
        for k in INT_KEYS:
            np.testing.assert_array_equal(dd[k].cpu().numpy(), want[k], err_msg=k)
        same = dd["aggregated_vote_inds"].cpu().numpy() == want["aggregated_vote_inds"]  # (B, num_proposal)
        assert same.mean() >= 0.95, same.mean()
        for k, v in want.items():
            if k in INT_KEYS or k == "aggregated_vote_inds":
                continue
            got = dd[k].detach().float().cpu().numpy()
            if k in PROPOSAL_KEYS:
                assert got.shape == v.shape, k  # proposal tensors are below the sub-sampling threshold
                got, v = got[same], v[same]
                if k in ("bbox_mask", "bbox_sems"):  # argmax of near-tied logits may flip within tolerance
                    assert (got == v).mean() > 0.98, k
                    continue
                np.testing.assert_allclose(got, v, rtol=rtol, atol=atol, err_msg=k)
                continue
            close(dd[k], v, rtol, atol)


def test_backbone_voting_proposal_c1_vs_reference_golden(oracle_backend, golden):
    """BASELINE config 1: 1 scene x 4096 pts, backbone + voting (+ proposal), no GPU."""
    run_c1(golden("pn2_backbone_c1.npz"), torch.device("cpu"), 2e-4, 2e-4)


def test_unused_reference_options_fail_loudly():
    from bridgeqa_amd.pointnet2_modules import PointnetSAModuleVotes
    from bridgeqa_amd.pointnet2_utils import QueryAndGroup
    with pytest.raises(NotImplementedError):
        QueryAndGroup(0.2, 16, sample_uniformly=True)
    with pytest.raises(NotImplementedError):
        PointnetSAModuleVotes(mlp=[3, 8], npoint=4, radius=0.2, nsample=4, pooling="avg")


def test_detector_weight_gradient_piece_count_host_logic():
    """pytorch_utils._wgrad_pieces: ~384 workgroups per launch, a multiple of 8 pieces (the kernel's XCD-aware order),
    never more pieces than 64-row K tiles"""
    from bridgeqa_amd.pytorch_utils import _wgrad_pieces, _WGRAD_WGS
    assert _WGRAD_WGS == 384
    for R, tiles in ((2097152, 3), (2097152, 1), (524288, 8), (131072, 10), (8192, 16), (300, 4), (64, 1)):
        ks = _wgrad_pieces(R, tiles)
        assert 1 <= ks <= (R + 63) // 64
        assert ks * tiles <= max(_WGRAD_WGS, tiles)
        assert ks < 8 or ks % 8 == 0
    assert _wgrad_pieces(2097152, 3) == 128 and _wgrad_pieces(300, 4) == 5 and _wgrad_pieces(64, 1) == 1

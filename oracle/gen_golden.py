"""Generate tests/golden/*.npz by running the REFERENCE's own Python layers in this container.

TEST INFRASTRUCTURE (build container only; /root/reference does not exist on the GPU box and
nothing under tests/ or bench.py reads it at run time).  Usage:

    python oracle/gen_golden.py            # writes tests/golden/pn2_*.npz

How the reference is run: `lib/pointnet2/pointnet2_utils.py:25-33` imports `pointnet2._ext`
(CUDA-only, unbuildable here) -- we pre-populate `sys.modules["pointnet2._ext"]` with the CPU
oracle (oracle/pn2_oracle.py) so that the reference's autograd Functions, QueryAndGroup,
PointnetSAModuleVotes, PointnetFPModule, Pointnet2Backbone, VotingModule and ProposalModule
execute unmodified on CPU.  Two further stand-ins, both for things ABSENT from /root/reference:
  * `data.scannet.model_util_scannet.ScannetDatasetConfig` (`data` is a dangling symlink): the
    VoteNet/ScanRefer class -- 18 classes, 1 heading bin, 18 size clusters, axis-aligned boxes
    (class2angle == 0), `mean_size_arr` from an absent .npz => a seeded synthetic array;
  * `.cuda()` on tensors (models/proposal_module.py:106,141) is made the identity.
Only inputs, outputs and state dicts (data) are written -- never reference source text.
"""
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")


def install_reference():
    sys.path.insert(0, REPO)
    from oracle import pn2_oracle

    pkg = types.ModuleType("pointnet2")
    pkg._ext = pn2_oracle
    sys.modules["pointnet2"] = pkg
    sys.modules["pointnet2._ext"] = pn2_oracle

    class ScannetDatasetConfig(object):
        """Stand-in for the absent data/scannet/model_util_scannet.py (VoteNet/ScanRefer)."""

        def __init__(self):
            self.num_class = 18
            self.num_heading_bin = 1
            self.num_size_cluster = 18
            rng = np.random.RandomState(7)
            self.mean_size_arr = rng.uniform(0.3, 2.0, size=(18, 3))

        def class2angle_batch(self, pred_cls, residual, to_label_format=True):
            return np.zeros(pred_cls.shape[0])

        def class2size_batch(self, pred_cls, residual):
            return self.mean_size_arr[pred_cls] + residual

        def param2obb_batch(self, center, heading_class, heading_residual, size_class, size_residual):
            heading_angle = self.class2angle_batch(heading_class, heading_residual)
            box_size = self.class2size_batch(size_class, size_residual)
            obb = np.zeros((heading_class.shape[0], 7))
            obb[:, 0:3] = center
            obb[:, 3:6] = box_size
            obb[:, 6] = heading_angle * -1
            return obb

    data = types.ModuleType("data")
    data.__path__ = []
    scannet = types.ModuleType("data.scannet")
    scannet.__path__ = []
    mus = types.ModuleType("data.scannet.model_util_scannet")
    mus.ScannetDatasetConfig = ScannetDatasetConfig
    sys.modules["data"] = data
    sys.modules["data.scannet"] = scannet
    sys.modules["data.scannet.model_util_scannet"] = mus

    torch.Tensor.cuda = lambda self, *a, **k: self  # CPU run of hard .cuda() calls

    os.chdir(REF)  # backbone_module.py:8 appends os.getcwd()+"/lib"
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "lib"))
    sys.path.insert(0, os.path.join(REF, "lib", "pointnet2"))
    return ScannetDatasetConfig


def npy(d):
    from tests.golden_util import subsample
    out = {}
    for k, v in d.items():
        if isinstance(v, torch.Tensor):
            out[k] = subsample(v.detach().cpu().numpy())
        else:
            out[k] = np.asarray(v)
    return out


def keys_of(prefix, module):
    """Fill the module deterministically by key name; record the key set (not the values)."""
    from tests.golden_util import fill_params
    return np.array(["%s %s" % (k, "x".join(map(str, s))) for k, s in fill_params(module, prefix)])


def scene(B, N, C, seed, room=(8.0, 8.0, 3.0)):
    g = torch.Generator().manual_seed(seed)
    xyz = torch.rand(B, N, 3, generator=g) * torch.tensor(room)
    if C == 0:
        return xyz.contiguous()
    feat = torch.randn(B, N, C, generator=g)
    return torch.cat([xyz, feat], -1).contiguous()


def main():
    DC = install_reference()
    os.makedirs(OUT, exist_ok=True)
    import pointnet2_utils as ref_utils  # reference lib/pointnet2/pointnet2_utils.py
    from lib.pointnet2.pointnet2_modules import PointnetSAModuleVotes, PointnetFPModule
    from models.backbone_module import Pointnet2Backbone
    from models.voting_module import VotingModule
    from models.proposal_module import ProposalModule

    # ---------------- operator level, through the reference's autograd Functions -------------
    ops = {}
    for tag, N, m in (("n64", 64, 16), ("n1000", 1000, 128), ("n4096", 4096, 512)):
        xyz = scene(2, N, 0, 100 + N)
        if N == 1000:  # duplicates, points inside the 1e-3 origin ball, exact ties
            xyz[0, 10] = xyz[0, 3]
            xyz[0, 500] = xyz[0, 3]
            xyz[0, 20] = torch.tensor([0.01, 0.02, 0.005])
            xyz[1, 0] = torch.tensor([0.0, 0.0, 0.0])
            xyz[1, 77] = torch.tensor([0.03, 0.0, 0.0])  # mag 9e-4 <= 1e-3 -> skipped
            xyz[1, 78] = torch.tensor([0.032, 0.0, 0.0])  # mag 1.024e-3 -> kept
            xyz[1, 100:108] = torch.tensor([4.0, 4.0, 1.5])  # 8 identical points
        inds = ref_utils.furthest_point_sample(xyz, m)
        ops["fps_%s_xyz" % tag] = xyz
        ops["fps_%s_m" % tag] = m
        ops["fps_%s_inds" % tag] = inds
        new_xyz = ref_utils.gather_operation(xyz.transpose(1, 2).contiguous(), inds).transpose(1, 2).contiguous()
        ops["gather_%s_out" % tag] = new_xyz
        for r, S in ((0.2, 8), (0.8, 16), (3.0, 32)):
            idx = ref_utils.ball_query(r, S, xyz, new_xyz)
            ops["bq_%s_r%g_S%d" % (tag, r, S)] = idx
    # group / interpolate with grads
    g = torch.Generator().manual_seed(5)
    xyz = scene(2, 512, 0, 11)
    inds = ref_utils.furthest_point_sample(xyz, 64)
    new_xyz = ref_utils.gather_operation(xyz.transpose(1, 2).contiguous(), inds).transpose(1, 2).contiguous()
    idx = ref_utils.ball_query(0.9, 16, xyz, new_xyz)
    feats = torch.randn(2, 7, 512, generator=g, requires_grad=True)
    grouped = ref_utils.grouping_operation(feats, idx)
    go = torch.randn(grouped.shape, generator=g)
    grouped.backward(go)
    ops.update(grp_xyz=xyz, grp_idx=idx, grp_feats=feats, grp_out=grouped, grp_go=go, grp_grad=feats.grad)
    # gather grad (vote-agg case: xyz requires grad)
    pts = torch.randn(2, 3, 512, generator=g, requires_grad=True)
    gout = ref_utils.gather_operation(pts, inds)
    ggo = torch.randn(gout.shape, generator=g)
    gout.backward(ggo)
    ops.update(gat_pts=pts, gat_inds=inds, gat_out=gout, gat_go=ggo, gat_grad=pts.grad)
    # three_nn / three_interpolate
    unknown = scene(2, 200, 0, 12)
    known = unknown[:, :50].clone().contiguous() + 0.0  # FPS-prefix-like: exact zero distances
    known[:, 25:] += 0.05
    dist, idx3 = ref_utils.three_nn(unknown, known)
    dist_recip = 1.0 / (dist + 1e-8)
    weight = dist_recip / torch.sum(dist_recip, dim=2, keepdim=True)
    kf = torch.randn(2, 9, 50, generator=g, requires_grad=True)
    interp = ref_utils.three_interpolate(kf, idx3, weight)
    igo = torch.randn(interp.shape, generator=g)
    interp.backward(igo)
    ops.update(nn_unknown=unknown, nn_known=known, nn_dist=dist, nn_idx=idx3, it_weight=weight,
               it_feats=kf, it_out=interp, it_go=igo, it_grad=kf.grad)
    np.savez_compressed(os.path.join(OUT, "pn2_ops.npz"), **npy(ops))

    # ---------------- module level ---------------------------------------------------------
    torch.manual_seed(0)
    mods = {}
    # SA module, train-mode BN (batch statistics) with grads, then eval mode
    sa = PointnetSAModuleVotes(npoint=64, radius=0.9, nsample=16, mlp=[5, 16, 16, 32], use_xyz=True, normalize_xyz=True)
    mods["sa_keys"] = keys_of("sa.", sa)
    pc = scene(2, 512, 5, 21)
    xyz = pc[..., :3].contiguous()
    feat = pc[..., 3:].transpose(1, 2).contiguous().requires_grad_(True)
    sa.train()
    nx, nf, ni = sa(xyz, feat)
    w = torch.randn(nf.shape, generator=g)
    (nf * w).sum().backward()
    mods.update(sa_pc=pc, sa_train_new_xyz=nx, sa_train_new_features=nf, sa_inds=ni, sa_w=w,
                sa_train_grad_features=feat.grad,
                sa_train_grad_w0=sa.mlp_module.layer0.conv.weight.grad,
                sa_train_running_mean2=sa.mlp_module.layer2.bn.bn.running_mean.clone(),
                sa_train_running_var2=sa.mlp_module.layer2.bn.bn.running_var.clone())
    sa.eval()
    nx, nf, ni = sa(xyz, feat.detach())
    mods.update(sa_eval_new_features=nf)
    # FP module
    fp = PointnetFPModule(mlp=[9 + 6, 16, 12])
    mods["fp_keys"] = keys_of("fp.", fp)
    fp.eval()
    uf = torch.randn(2, 6, 200, generator=g)
    out = fp(unknown, known, uf, kf.detach())
    mods.update(fp_unknown=unknown, fp_known=known, fp_unknown_feats=uf, fp_known_feats=kf.detach(), fp_out=out)
    np.savez_compressed(os.path.join(OUT, "pn2_modules.npz"), **npy(mods))

    # ---------------- BASELINE config 1: backbone + voting (+ proposal) on 1 x 4096 x (3+1) ---
    torch.manual_seed(0)
    bb = Pointnet2Backbone(input_feature_dim=1)
    vote = VotingModule(1, 256)
    dc = DC()
    prop = ProposalModule(dc.num_class, dc.num_heading_bin, dc.num_size_cluster, dc.mean_size_arr, 256, "vote_fps")
    pc = scene(1, 4096, 1, 42)
    c1 = {"point_clouds": pc, "mean_size_arr": dc.mean_size_arr}
    c1["keys"] = np.concatenate([keys_of("detection_backbone.", bb), keys_of("voting_net.", vote),
                                 keys_of("proposal_net.", prop)])
    for mode in ("eval", "train"):
        bb.train(mode == "train"); vote.train(mode == "train"); prop.train(mode == "train")
        # weights are filled BEFORE any forward; the eval pass runs first and leaves BN stats untouched
        dd = bb({"point_clouds": pc})
        vx, vf = vote(dd["fp2_xyz"], dd["fp2_features"])
        fnorm = torch.norm(vf, p=2, dim=1)  # qa_module.py:452-453
        vf = vf.div(fnorm.unsqueeze(1))
        dd["vote_xyz"], dd["vote_features"] = vx, vf
        dd = prop(vx, vf, dd)
        for k, v in dd.items():
            if k != "point_clouds":
                c1["%s.%s" % (mode, k)] = v
    np.savez_compressed(os.path.join(OUT, "pn2_backbone_c1.npz"), **npy(c1))
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()

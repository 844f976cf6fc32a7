"""Golden vectors for the detection losses, produced by the REFERENCE's own lib/loss_helper.py in this build container.

TEST INFRASTRUCTURE.  Usage:  python oracle/gen_golden_loss.py   -> tests/golden/det_loss.npz

Import-time shims, all for names OUTSIDE the reference tree or for the missing GPU:
  * icecream, trimesh, plyfile                  -> empty modules (never called on this path)
  * data.scannet.model_util_scannet             -> absent from the reference checkout (ScanNet tooling); empty module
  * torch.Tensor.cuda -> identity, torch.cuda.FloatTensor -> torch.FloatTensor: the reference hard-codes .cuda()
    (loss_helper.py:98-99,160,175,183); on this GPU-less container the same arithmetic runs on the CPU.
Inputs are seeded random tensors with the shapes of SURVEY.md §8d (small B / K); the file stores inputs and the
reference's outputs (the three loss functions + the weighted x10 total of get_loss with detection only)."""
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden", "det_loss.npz")


def install_shims():
    for name in ("icecream", "trimesh", "plyfile", "data", "data.scannet", "data.scannet.model_util_scannet"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["icecream"].ic = lambda *a, **k: None
    sys.modules["plyfile"].PlyData = object
    sys.modules["plyfile"].PlyElement = object
    sys.modules["data.scannet.model_util_scannet"].extract_pc_in_box3d = None
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda.FloatTensor = torch.FloatTensor
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "lib"))


def make_inputs(seed=0, B=3, N=600, num_seed=64, K=32, K2=10, n_valid=6, NH=1, NS=18, NC=18):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.rand(*s, generator=g)
    n = lambda *s: torch.randn(*s, generator=g)
    d = {}
    d["seed_xyz"] = r(B, num_seed, 3) * 4
    d["vote_xyz"] = d["seed_xyz"] + n(B, num_seed, 3) * 0.3
    d["seed_inds"] = torch.stack([torch.randperm(N, generator=g)[:num_seed] for _ in range(B)]).int()
    d["vote_label"] = n(B, N, 9) * 0.5
    d["vote_label_mask"] = (r(B, N) < 0.6).long()
    d["center_label"] = r(B, K2, 3) * 4
    d["box_label_mask"] = torch.zeros(B, K2)
    d["box_label_mask"][:, :n_valid] = 1
    # proposals: some near GT centres, some far, some in the grey zone
    near = d["center_label"][:, torch.randint(0, n_valid, (K,), generator=g)] + n(B, K, 3) * 0.25
    d["aggregated_vote_xyz"] = near
    d["center"] = near + n(B, K, 3) * 0.1
    d["objectness_scores"] = n(B, K, 2)
    d["heading_scores"] = n(B, K, NH)
    d["heading_residuals_normalized"] = n(B, K, NH) * 0.2
    d["size_scores"] = n(B, K, NS)
    d["size_residuals_normalized"] = n(B, K, NS, 3) * 0.2
    d["sem_cls_scores"] = n(B, K, NC)
    d["heading_class_label"] = torch.randint(0, NH, (B, K2), generator=g)
    d["heading_residual_label"] = n(B, K2) * 0.1
    d["size_class_label"] = torch.randint(0, NS, (B, K2), generator=g)
    d["size_residual_label"] = n(B, K2, 3) * 0.1
    d["sem_cls_label"] = torch.randint(0, NC, (B, K2), generator=g)
    mean_size_arr = (r(NS, 3) + 0.5).numpy().astype(np.float64)
    return d, mean_size_arr, dict(NH=NH, NS=NS, NC=NC)


def main():
    install_shims()
    import lib.loss_helper as lh
    d, mean_size_arr, dims = make_inputs()
    cfg = types.SimpleNamespace(num_heading_bin=dims["NH"], num_size_cluster=dims["NS"], num_class=dims["NC"],
                                mean_size_arr=mean_size_arr)
    inputs = {k: v.clone() for k, v in d.items()}
    out = {}
    out["vote_loss"] = lh.compute_vote_loss(d)
    ol, lab, mask, assign = lh.compute_objectness_loss(d)
    out.update(objectness_loss=ol, objectness_label=lab, objectness_mask=mask, object_assignment=assign)
    d["objectness_label"], d["objectness_mask"], d["object_assignment"] = lab, mask, assign
    names = ("center_loss", "heading_cls_loss", "heading_reg_loss", "size_cls_loss", "size_reg_loss", "sem_cls_loss")
    for k, v in zip(names, lh.compute_box_and_sem_cls_loss(d, cfg)):
        out[k] = v
    # get_loss with detection only, scripts/train.py weights (vote 1.0, objectness 0.5, box 1.0, sem-cls 0.1), x10
    box = out["center_loss"] + 0.1 * out["heading_cls_loss"] + out["heading_reg_loss"] + 0.1 * out["size_cls_loss"] \
        + out["size_reg_loss"]
    out["total_x10"] = 10 * (1.0 * out["vote_loss"] + 0.5 * out["objectness_loss"] + 1.0 * box + 0.1 * out["sem_cls_loss"])
    save = {"in_" + k: v.numpy() for k, v in inputs.items()}
    save.update({"out_" + k: v.detach().numpy() for k, v in out.items()})
    save["mean_size_arr"] = mean_size_arr
    save["dims"] = np.array([dims["NH"], dims["NS"], dims["NC"]])
    np.savez_compressed(OUT, **save)
    print("wrote", OUT, {k: float(v) for k, v in out.items() if v.dim() == 0})


if __name__ == "__main__":
    main()

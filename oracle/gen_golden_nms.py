"""Golden vectors for the proposal post-processing, produced by the REFERENCE's own
lib/ap_helper.py parse_predictions (:40-178) and utils/nms.py in this build container.

TEST INFRASTRUCTURE.  Usage:  python oracle/gen_golden_nms.py   -> tests/golden/nms.npz

Shims (as the other generators): icecream / trimesh / plyfile -> empty modules; data.scannet.model_util_scannet is ABSENT
from the reference checkout: `extract_pc_in_box3d` is restated from VoteNet's published model_util (scipy Delaunay hull of
the 8 corners, find_simplex >= 0) and the dataset config (class2angle / class2size, num_class, mean_size_arr) from its
ScannetDatasetConfig.  Inputs are seeded random proposals clustered around a few centres (so that boxes overlap), stored
with the reference's outputs for five config_dict variants."""
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden", "nms.npz")
sys.path.insert(0, os.path.join(REPO, "oracle"))
from gen_golden_loss import install_shims  # noqa: E402


def in_hull(p, hull):
    from scipy.spatial import Delaunay
    if not isinstance(hull, Delaunay):
        hull = Delaunay(hull)
    return hull.find_simplex(p) >= 0


def extract_pc_in_box3d(pc, box3d):
    inds = in_hull(pc[:, 0:3], box3d)
    return pc[inds, :], inds


class ConfigShim(object):
    def __init__(self, num_heading_bin, num_class, mean_size_arr):
        self.num_heading_bin, self.num_class, self.mean_size_arr = num_heading_bin, num_class, mean_size_arr

    def class2angle(self, pred_cls, residual, to_label_format=True):
        angle = pred_cls * (2 * np.pi / float(self.num_heading_bin)) + residual
        if to_label_format and angle > np.pi:
            angle = angle - 2 * np.pi
        return angle

    def class2size(self, pred_cls, residual):
        return self.mean_size_arr[int(pred_cls), :] + residual


VARIANTS = (
    dict(remove_empty_box=True, use_3d_nms=True, nms_iou=0.25, use_old_type_nms=False, cls_nms=True, per_class_proposal=True, conf_thresh=0.05),
    dict(remove_empty_box=False, use_3d_nms=True, nms_iou=0.25, use_old_type_nms=False, cls_nms=False, per_class_proposal=False, conf_thresh=0.3),
    dict(remove_empty_box=False, use_3d_nms=False, nms_iou=0.3, use_old_type_nms=False, cls_nms=False, per_class_proposal=False, conf_thresh=0.05),
    dict(remove_empty_box=True, use_3d_nms=True, nms_iou=0.2, use_old_type_nms=True, cls_nms=False, per_class_proposal=False, conf_thresh=0.05),
    dict(remove_empty_box=False, use_3d_nms=False, nms_iou=0.25, use_old_type_nms=True, cls_nms=False, per_class_proposal=True, conf_thresh=0.5),
)


def make_inputs(seed, NH, B=2, K=48, N=3000, NS=18, NC=18):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.rand(*s, generator=g)
    n = lambda *s: torch.randn(*s, generator=g)
    hubs = r(B, 6, 3) * torch.tensor([6.0, 6.0, 2.5])
    d = {}
    d["center"] = hubs[:, torch.randint(0, 6, (K,), generator=g)] + n(B, K, 3) * 0.25
    d["heading_scores"] = n(B, K, NH)
    d["heading_residuals"] = n(B, K, NH) * 0.3
    d["size_scores"] = n(B, K, NS)
    mean_size_arr = (r(NS, 3) * 0.8 + 0.4).numpy().astype(np.float64)
    d["size_residuals"] = n(B, K, NS, 3) * 0.05
    d["sem_cls_scores"] = n(B, K, NC)
    d["objectness_scores"] = n(B, K, 2) * 2.0
    pc = r(B, N, 3) * torch.tensor([6.5, 6.5, 3.0])
    d["point_clouds"] = torch.cat([pc, n(B, N, 2)], -1)       # xyz + 2 feature channels
    return d, mean_size_arr


def main():
    install_shims()
    sys.modules["data.scannet.model_util_scannet"].extract_pc_in_box3d = extract_pc_in_box3d
    import lib.ap_helper as ap
    save = {}
    for v, cfgd in enumerate(VARIANTS):
        NH = 1 if v % 2 == 0 else 4
        d, mean_size_arr = make_inputs(10 + v, NH)
        for k, t in d.items():
            save["v%d_in_%s" % (v, k)] = t.numpy()
        save["v%d_mean_size_arr" % v] = mean_size_arr
        save["v%d_NH" % v] = np.array(NH)
        cfgd = dict(cfgd, dataset_config=ConfigShim(NH, 18, mean_size_arr))
        end = {k: t.clone() for k, t in d.items()}
        out = ap.parse_predictions(end, cfgd)
        save["v%d_pred_mask" % v] = np.asarray(end["pred_mask"])
        for i, lst in enumerate(out):
            save["v%d_b%d_cls" % (v, i)] = np.array([c for c, _, _ in lst], dtype=np.int64)
            save["v%d_b%d_score" % (v, i)] = np.array([s for _, _, s in lst], dtype=np.float64)
            save["v%d_b%d_corners" % (v, i)] = np.array([c for _, c, _ in lst], dtype=np.float64).reshape(-1, 8, 3)
        print("variant", v, "kept per scene", [int(m.sum()) for m in np.asarray(end["pred_mask"])], "listed", [len(l) for l in out])
    np.savez_compressed(OUT, **save)
    print("wrote", OUT, os.path.getsize(OUT))


if __name__ == "__main__":
    main()

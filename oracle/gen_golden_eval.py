"""Golden vectors for the per-iteration evaluation of the training loop (lib/solver.py:542 calls
lib/eval_helper.py get_eval (:57-320) after EVERY forward, train and val), produced by the REFERENCE's own Python in this
build container: get_loss (lib/loss_helper.py:355-464) first, as solver.py:527-541 does, then get_eval on the same dict.

TEST INFRASTRUCTURE.  Usage:  python oracle/gen_golden_eval.py   -> tests/golden/eval.npz

Variants: 0 = plain (lang classifier on, all four answer-score branches present); 1 = use_oracle, no lang classifier, only
`answer_scores`; 2 = use_best; 3 = post_processing (parse_predictions: remove_empty_box + per-class 3-D NMS) ahead of the
masks; 4 = use_best where the best proposal is often NOT predicted to be an object (the masked-else quirk).  use_cat_rand
draws from torch.randperm and has no golden.

Shims (as oracle/gen_golden_qa.py / gen_golden_nms.py): icecream / trimesh / plyfile -> empty modules;
data.scannet.model_util_scannet is ABSENT from the reference checkout -- the ScannetDatasetConfig members get_loss,
get_eval and parse_predictions call (param2obb[_batch], class2angle[_batch], class2size[_batch]) and extract_pc_in_box3d
are restated from VoteNet / ScanRefer's published model_util; torch.Tensor.cuda -> identity.
"""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, "tests", "golden", "eval.npz")
sys.path.insert(0, os.path.join(REPO, "oracle"))

from gen_golden_loss import install_shims  # noqa: E402
from gen_golden_nms import extract_pc_in_box3d  # noqa: E402
from gen_golden_qa import DatasetConfigShim, loss_inputs  # noqa: E402


class EvalConfigShim(DatasetConfigShim):
    """+ the scalar members (ScannetDatasetConfig.class2angle / class2size / param2obb)"""

    def class2angle(self, pred_cls, residual, to_label_format=True):
        angle = pred_cls * (2 * np.pi / float(self.num_heading_bin)) + residual
        if to_label_format and angle > np.pi:
            angle = angle - 2 * np.pi
        return angle

    def class2size(self, pred_cls, residual):
        return self.mean_size_arr[int(pred_cls), :] + residual

    def param2obb(self, center, heading_class, heading_residual, size_class, size_residual):
        obb = np.zeros((7,))
        obb[0:3] = center
        obb[3:6] = self.class2size(int(size_class), size_residual)
        obb[6] = self.class2angle(heading_class, heading_residual) * -1
        return obb


VARIANTS = (
    dict(seed=5, NH=1, kw=dict(use_reference=True, use_lang_classifier=True), extra=("scene", "2d", "2d3d")),
    dict(seed=6, NH=4, kw=dict(use_reference=True, use_lang_classifier=False, use_oracle=True), extra=()),
    dict(seed=7, NH=1, kw=dict(use_reference=True, use_lang_classifier=True, use_best=True), extra=("scene", "2d")),
    dict(seed=8, NH=4, kw=dict(use_reference=True, use_lang_classifier=True), extra=("2d3d",),
         post=dict(remove_empty_box=True, use_3d_nms=True, nms_iou=0.25, use_old_type_nms=False, cls_nms=True,
                   per_class_proposal=True, conf_thresh=0.05)),
    # use_best with most proposals predicted "not an object": the reference's `if use_best` / separate `if use_cat_rand ...
    # else` then masks the replaced cluster_ref (eval_helper.py:97-124), which differs from the unmasked labels
    dict(seed=9, NH=1, kw=dict(use_reference=True, use_lang_classifier=True, use_best=True), extra=(), obj_bias=-0.9),
)
POST_KEYS = ("remove_empty_box", "use_3d_nms", "nms_iou", "use_old_type_nms", "cls_nms", "per_class_proposal", "conf_thresh")
EVAL_INPUTS = ("objectness_scores", "objectness_label", "objectness_mask", "object_assignment", "cluster_ref", "cluster_labels",
               "center", "heading_scores", "heading_residuals", "size_scores", "size_residuals", "sem_cls_scores",
               "center_label", "heading_class_label", "heading_residual_label", "size_class_label", "size_residual_label",
               "sem_cls_label", "ref_box_label", "lang_scores", "object_cat", "answer_scores", "answer_scores_scene",
               "answer_scores_2d", "answer_scores_2d3d", "answer_cats", "point_clouds")
OUT_TENSORS = ("lang_acc", "answer_acc_at1", "answer_acc_at10", "answer_acc_at1_scene", "answer_acc_at10_scene",
               "answer_acc_at1_2d", "answer_acc_at10_2d", "answer_acc_at1_2d3d", "answer_acc_at10_2d3d",
               "answer_acc_at1_3d_over_2d", "answer_acc_at1_2d_over_3d", "obj_acc", "sem_acc", "pred_mask", "label_mask",
               "cluster_ref", "cluster_labels", "pred_center", "pred_heading_class", "pred_heading_residual",
               "pred_size_class", "pred_size_residual", "pred_answers", "pred_answers_at10")


def main():
    install_shims()
    sys.modules["data.scannet.model_util_scannet"].extract_pc_in_box3d = extract_pc_in_box3d
    import lib.loss_helper as lh
    import lib.eval_helper as eh
    weights = dict(vote_loss=1.0, objectness_loss=0.5, box_loss=1.0, sem_cls_loss=0.1, ref_loss=0.1, lang_loss=0.1,
                   answer_loss=1.0)
    save = {"post_keys": np.array(POST_KEYS)}
    for v, var in enumerate(VARIANTS):
        d, mean_size_arr, dims = loss_inputs(var["seed"], var["NH"])
        g = torch.Generator().manual_seed(2000 + var["seed"])
        B, K = d["center"].shape[:2]
        K2, A = d["center_label"].shape[1], d["answer_scores"].shape[1]
        d["answer_scores"] = torch.randn(B, A, generator=g)
        for name in ("scene", "2d", "2d3d"):
            d.pop("answer_scores_" + name, None)
            if name in var["extra"]:
                d["answer_scores_" + name] = d["answer_scores"] + torch.randn(B, A, generator=g) * 0.7
        cats = (torch.rand(B, A, generator=g) < 0.1).float()
        cats[torch.arange(B), d["answer_cat"]] = 1.0
        d["answer_cats"] = cats
        # the referred object = the GT box loss_inputs drew ref_center_label from; reference scores that favour proposals
        # near it, so that some of the predicted boxes overlap the referred one
        ref_idx = (d["center_label"] - d["ref_center_label"][:, None]).norm(dim=-1).argmin(1)
        d["ref_box_label"] = torch.zeros(B, K2, dtype=torch.long)
        d["ref_box_label"][torch.arange(B), ref_idx] = 1
        dist = (d["center"] - d["ref_center_label"][:, None]).norm(dim=-1)
        d["cluster_ref"] = torch.randn(B, K, generator=g) + 4.0 * torch.exp(-dist * dist / 0.05)
        d["point_clouds"] = torch.cat([torch.rand(B, 2500, 3, generator=g) * 4.2 - 0.1, torch.randn(B, 2500, 1, generator=g)], -1)
        # objectness logits that favour "object" near GT centres, so that the masks are not trivially empty / full
        d["objectness_scores"] = d["objectness_scores"] + torch.tensor([0.0, var.get("obj_bias", 0.6)])
        cfg = EvalConfigShim(dims["NH"], dims["NS"], dims["NC"], mean_size_arr)
        _, d = lh.get_loss(d, cfg, detection=True, use_reference=True, use_lang_classifier=True, use_answer=True,
                           loss_weights=weights)
        d = {k: (t.detach() if torch.is_tensor(t) else t) for k, t in d.items()}
        for k, t in d.items():
            if torch.is_tensor(t) and k in EVAL_INPUTS and (k != "point_clouds" or "post" in var):
                save["v%d_in_%s" % (v, k)] = t.clone().numpy()
        save["v%d_mean_size_arr" % v] = mean_size_arr
        save["v%d_dims" % v] = np.array([dims["NH"], dims["NS"], dims["NC"]])
        kw = dict(var["kw"])
        save["v%d_flags" % v] = np.array([int(kw.get(k, False)) for k in ("use_reference", "use_lang_classifier", "use_oracle",
                                                                           "use_cat_rand", "use_best")])
        if "post" in var:
            kw["post_processing"] = dict(var["post"], dataset_config=cfg)
            save["v%d_post" % v] = np.array([float(var["post"][k]) for k in POST_KEYS])
        out = eh.get_eval(d, cfg, **kw)
        for k in OUT_TENSORS:
            if k in out:
                save["v%d_out_%s" % (v, k)] = np.asarray(out[k].detach().numpy() if torch.is_tensor(out[k]) else out[k])
        save["v%d_out_ref_acc" % v] = np.asarray(out["ref_acc"], dtype=np.float64)
        save["v%d_out_ref_iou" % v] = np.asarray(out["ref_iou"], dtype=np.float64)
        save["v%d_out_ref_iou_rates" % v] = np.array([out["ref_iou_rate_0.25"], out["ref_iou_rate_0.5"]], dtype=np.float64)
        save["v%d_out_pred_bboxes" % v] = np.asarray(out["pred_bboxes"], dtype=np.float64)
        save["v%d_out_gt_bboxes" % v] = np.asarray(out["gt_bboxes"], dtype=np.float64)
        if "pred_langs" in out:
            save["v%d_out_pred_langs" % v] = out["pred_langs"]
        if var["kw"].get("use_best"):
            lab = torch.as_tensor(save["v%d_out_cluster_labels" % v])
            print("  use_best: samples whose cluster_ref differs from the unmasked labels:",
                  int((torch.as_tensor(save["v%d_out_cluster_ref" % v]) != lab).any(1).sum()), "of", lab.shape[0])
        print("variant", v, "ref_acc", out["ref_acc"], "iou", np.round(out["ref_iou"], 3).tolist(),
              "rates", out["ref_iou_rate_0.25"], out["ref_iou_rate_0.5"], "pred_mask sum", float(out["pred_mask"].sum()),
              "acc@1", float(out["answer_acc_at1"]), "acc@10", float(out["answer_acc_at10"]), "obj_acc", float(out["obj_acc"]),
              "sem_acc", float(out["sem_acc"]))
    np.savez_compressed(OUT, **save)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()

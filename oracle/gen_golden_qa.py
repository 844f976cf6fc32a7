"""Golden vectors for the rest of the VQA branch, produced by the REFERENCE's own Python in this
build container:

  A. lib/loss_helper.py get_loss (:355-464) with every term on -- compute_reference_loss (:196-244, the numpy IoU loop),
     compute_lang_classification_loss (:247-251), compute_answer_classification_loss (:254-279), lib/loss.py
     SoftmaxRankingLoss -- in two variants: answer loss from the text decoder (`decoder_loss`) and from the closed answer
     set (+ the scene-branch scores);
  B. the heads that follow the BLIP fusion in ScanQA.forward (models/qa_module.py:735-754) built from the reference's own
     models/mcan_module.py classes (SGA, and SA for the state-dict key list) and the nn.Sequential heads of
     qa_module.py:234-249, run in the order of :735-754 (the statements are quoted in `heads_reference`).  ScanQA itself is
     not instantiated: its constructor loads BLIP / BERT checkpoints from the network.

TEST INFRASTRUCTURE.  Usage:  python oracle/gen_golden_qa.py   -> tests/golden/qa_heads.npz

Import-time shims (as oracle/gen_golden_loss.py): icecream / trimesh / plyfile -> empty modules;
data.scannet.model_util_scannet -> ABSENT from the reference checkout (ScanNet tooling): the dataset-config members
get_loss needs (param2obb_batch, class2angle_batch, class2size_batch, mean_size_arr, bin counts) are restated below from
VoteNet / ScanRefer's published ScannetDatasetConfig; torch.Tensor.cuda -> identity (the reference hard-codes .cuda()).
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden", "qa_heads.npz")
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, os.path.join(REPO, "oracle"))

from gen_golden_loss import install_shims, make_inputs  # noqa: E402
from golden_util import fill_params  # noqa: E402


class DatasetConfigShim(object):
    """the members of ScannetDatasetConfig (data/scannet/model_util_scannet.py of VoteNet / ScanRefer) that
    lib/loss_helper.py calls; numpy, float64, as upstream"""

    def __init__(self, num_heading_bin, num_size_cluster, num_class, mean_size_arr):
        self.num_heading_bin, self.num_size_cluster, self.num_class = num_heading_bin, num_size_cluster, num_class
        self.mean_size_arr = mean_size_arr

    def class2angle_batch(self, pred_cls, residual, to_label_format=True):
        angle = pred_cls * (2 * np.pi / float(self.num_heading_bin)) + residual
        if to_label_format:
            angle[angle > np.pi] = angle[angle > np.pi] - 2 * np.pi
        return angle

    def class2size_batch(self, pred_cls, residual):
        return self.mean_size_arr[pred_cls] + residual

    def param2obb_batch(self, center, heading_class, heading_residual, size_class, size_residual):
        obb = np.zeros((heading_class.shape[0], 7))
        obb[:, 0:3] = center
        obb[:, 3:6] = self.class2size_batch(size_class, size_residual)
        obb[:, 6] = self.class2angle_batch(heading_class, heading_residual) * -1
        return obb


def loss_inputs(seed, NH):
    d, mean_size_arr, dims = make_inputs(seed=seed, NH=NH)
    g = torch.Generator().manual_seed(1000 + seed)
    B, K = d["center"].shape[:2]
    K2 = d["center_label"].shape[1]
    NS, NC, A = dims["NS"], dims["NC"], 30
    msa = torch.from_numpy(mean_size_arr).float()
    d["heading_residuals"] = d["heading_residuals_normalized"] * (np.pi / NH)
    d["size_residuals"] = d["size_residuals_normalized"] * msa[None, None]
    d["cluster_ref"] = torch.randn(B, K, generator=g)
    pick = torch.randint(0, 6, (B,), generator=g)  # the referred object = one of the valid GT boxes
    bi = torch.arange(B)
    d["ref_center_label"] = d["center_label"][bi, pick]
    d["ref_heading_class_label"] = d["heading_class_label"][bi, pick]
    d["ref_heading_residual_label"] = d["heading_residual_label"][bi, pick]
    d["ref_size_class_label"] = d["size_class_label"][bi, pick]
    d["ref_size_residual_label"] = d["size_residual_label"][bi, pick]
    d["ref_obj_mask"] = torch.tensor([1.0, 0.0, 1.0])[:B]
    d["lang_scores"] = torch.randn(B, NC, generator=g)
    d["object_cat"] = torch.randint(0, NC, (B,), generator=g)
    d["answer_scores"] = torch.randn(B, A, generator=g)
    d["answer_scores_scene"] = torch.randn(B, A, generator=g)
    d["answer_cat"] = torch.randint(0, A, (B,), generator=g)
    return d, mean_size_arr, dims


def heads_reference(mods, data_dict, object_feat, object_mask, fused_feat, fused_mask):
    """models/qa_module.py:735-754, statement by statement, on the reference's module classes"""
    data_dict["lang_scores"] = mods.lang_cls(fused_feat[:, 0, :])                                   # :736
    fused_feat_for_crossatt = mods.linear_blip_to_object(fused_feat)                              # :746
    fused_mask = fused_mask.unsqueeze(1).unsqueeze(2).bool()                                      # :748
    for dec in mods.dec_list_qo:                                                                  # :749-750
        object_feat = dec(object_feat, fused_feat_for_crossatt, ~object_mask, ~fused_mask, att_pdrop=None,
                          att_drop_topk=None)
    object_conf_feat = object_feat * data_dict["objectness_scores"].max(2)[1].float().unsqueeze(2)   # :751-753
    data_dict["cluster_ref"] = mods.object_cls(object_conf_feat).squeeze(-1)                      # :754
    return data_dict, object_feat


def main():
    install_shims()
    import lib.loss_helper as lh
    from models.mcan_module import SA, SGA
    save = {}
    weights = dict(vote_loss=1.0, objectness_loss=0.5, box_loss=1.0, sem_cls_loss=0.1, ref_loss=0.1, lang_loss=0.1,
                   answer_loss=1.0)
    save["loss_weights"] = np.array([weights[k] for k in sorted(weights)])
    save["loss_weight_keys"] = np.array(sorted(weights))
    # ---- A. losses: variant 0 = closed-set answer scores (+ scene branch), NH = 1 (ScanNet); variant 1 = decoder loss, NH = 4
    for v, (seed, NH, with_decoder) in enumerate(((3, 1, False), (4, 4, True))):
        d, mean_size_arr, dims = loss_inputs(seed, NH)
        if with_decoder:
            d["decoder_loss"] = torch.tensor(2.375)
            del d["answer_scores_scene"]
        cfg = DatasetConfigShim(dims["NH"], dims["NS"], dims["NC"], mean_size_arr)
        for k, t in d.items():
            save["l%d_in_%s" % (v, k)] = t.clone().numpy()
        save["l%d_mean_size_arr" % v] = mean_size_arr
        save["l%d_dims" % v] = np.array([dims["NH"], dims["NS"], dims["NC"]])
        loss, out = lh.get_loss(d, cfg, detection=True, use_reference=True, use_lang_classifier=True, use_answer=True,
                                loss_weights=weights)
        for k in ("loss", "ref_loss", "lang_loss", "answer_loss", "cluster_labels", "vote_loss", "objectness_loss",
                  "box_loss", "sem_cls_loss"):
            save["l%d_out_%s" % (v, k)] = out[k].detach().numpy()
        print("variant", v, "loss", float(loss), "ref", float(out["ref_loss"]), "lang", float(out["lang_loss"]),
              "answer", float(out["answer_loss"]), "labels at", out["cluster_labels"].argmax(1).tolist())
    # ---- B. heads after the fusion
    hidden, enc, NC, layers, heads, pdrop = 64, 96, 18, 2, 4, 0.1
    mods = nn.Module()
    mods.enc_list_o = nn.ModuleList([SA(hidden, heads, pdrop) for _ in range(layers)])                      # qa_module.py:224
    mods.lang_cls = nn.Sequential(nn.Linear(enc, hidden), nn.GELU(), nn.Dropout(0.1), nn.Linear(hidden, NC))   # :234-239
    mods.object_cls = nn.Sequential(nn.Linear(hidden, hidden), nn.GELU(), nn.Dropout(0.1), nn.Linear(hidden, 1))  # :242-247
    mods.linear_blip_to_object = nn.Linear(enc, hidden)                                                    # :248
    mods.dec_list_qo = nn.ModuleList([SGA(hidden, heads, pdrop) for _ in range(layers)])                    # :249
    save["head_keys"] = np.array([k for k, _ in fill_params(mods, "qa.")])
    mods.eval()
    g = torch.Generator().manual_seed(77)
    B, K, L = 3, 24, 9
    object_feat = torch.randn(B, K, hidden, generator=g)
    fused_feat = torch.randn(B, L, enc, generator=g)
    objectness = torch.randn(B, K, 2, generator=g)
    bbox_mask = objectness.argmax(-1)
    object_mask = (~bbox_mask.bool()).unsqueeze(1).unsqueeze(2)          # qa_module.py:466-472: True = not an object
    fused_mask = torch.ones(B, L, dtype=torch.long)
    fused_mask[0, 6:] = 0
    fused_mask[2, 8:] = 0
    dd = {"objectness_scores": objectness}
    with torch.no_grad():
        dd, obj_out = heads_reference(mods, dd, object_feat, object_mask, fused_feat, fused_mask)
        sa_out = mods.enc_list_o[0](object_feat, object_mask)            # the SA block alone (never called by the reference)
    save.update(h_object_feat=object_feat.numpy(), h_fused_feat=fused_feat.numpy(), h_objectness=objectness.numpy(),
                h_fused_mask=fused_mask.numpy(), h_lang_scores=dd["lang_scores"].numpy(),
                h_cluster_ref=dd["cluster_ref"].numpy(), h_object_out=obj_out.numpy(), h_sa_out=sa_out.numpy(),
                h_dims=np.array([hidden, enc, NC, layers, heads]))
    np.savez_compressed(OUT, **save)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()

"""Per-iteration evaluation of the training loop -- mirror of the reference's lib/eval_helper.py `get_eval` (:57-320), which
lib/solver.py:542 calls after EVERY forward (train and val): SURVEY.md §8f rank 1, the last per-step host loop on the caller
side of the hot path.  Same function name, arguments and data_dict keys.

What changes underneath.  The reference decodes the referred box of each sample in a Python loop with ten
`.detach().cpu().numpy()` round trips per sample (eval_helper.py:171-197: 160 device->host synchronisations per step at
batch 16), builds the eight corners with numpy and reduces them to an axis-aligned IoU; `ref_acc`, `pred_langs` and the IoU
rates are further host copies.  Here the decode, the corners and the IoU are one batched computation on the device (fp64:
the reference's numpy arithmetic is double), the accuracies stay device scalars, and the host-format outputs the
reference's callers read (`ref_acc`, `ref_iou` lists, the two IoU rates, `pred_bboxes` / `gt_bboxes` corner arrays,
`pred_langs`) come from ONE packed device->host copy at the end -- or from none with `host_outputs=False`, where they
stay device tensors for solver.PackedRunningLog to fold into its single per-step copy.

`ScannetDatasetConfig` (data/scannet/model_util_scannet.py) is ABSENT from the reference checkout: param2obb / class2angle /
class2size are restated from VoteNet's published config (loss_helper._class2angle; size = mean_size_arr[class] + residual;
obb heading = -angle), reading num_heading_bin and mean_size_arr from `config`.

Reference quirks kept on purpose:
  * `cluster_labels *= label_masks` acts IN PLACE on data_dict["cluster_labels"] when that is fp32 (`.float()` of an fp32
    tensor is the tensor itself), so the masked labels are what `use_best` and later readers see (:88-89);
  * `ref_acc` divides by (1 + 1e-8) (:93-94); `sem_acc` divides by pred_mask.sum() without an epsilon (:317);
  * `use_reference` is accepted and not read (the reference never tests it inside get_eval);
  * `if use_best: ...` is followed by a SEPARATE `if use_cat_rand: ... else: ...` (:97-124): under use_best alone the else
    branch still runs on the replaced cluster_ref -- pred_ref = argmax(cluster_labels * pred_masks) and cluster_ref =
    cluster_labels * pred_masks, which differ from the unmasked labels whenever the best proposal is not predicted to be
    an object.
`use_cat_rand` (:103-121) draws ONE uniformly random proposal among those assigned to a GT box of the question's category
-- here the arg-max of uniform noise over the candidate mask (the same distribution, not the same random stream, and no
host synchronisation); with no candidate both forms pick proposal 0.
"""
import numpy as np
import torch
import torch.nn.functional as F

from .loss_helper import _aabb_of_obb, _class2angle, _const


def construct_bbox_corners(center, box_size):
    """eval_helper.py:41-54, batched on the device: (..., 3), (..., 3) -> (..., 8, 3), heading ignored as there"""
    sx, sy, sz = box_size[..., 0:1] / 2, box_size[..., 1:2] / 2, box_size[..., 2:3] / 2
    x = torch.cat([sx, sx, -sx, -sx, sx, sx, -sx, -sx], -1)
    y = torch.cat([sy, -sy, -sy, sy, sy, -sy, -sy, sy], -1)
    z = torch.cat([sz, sz, sz, sz, -sz, -sz, -sz, -sz], -1)
    return torch.stack([x, y, z], -1) + center.unsqueeze(-2)


def _answer_acc(scores, cats, topk=10):
    """eval_helper.py:214-229: top-1 / top-10 hit rates against the multi-hot `answer_cats`; -> acc1, acc10, pred1, pred10,
    per-sample top-1 hit mask"""
    num_classes = scores.shape[1]
    pred1 = torch.argmax(scores, 1)
    hit1 = (F.one_hot(pred1, num_classes=num_classes).float() * cats).max(dim=1)[0]
    pred10 = scores.topk(topk, dim=1)[1]
    hit10 = (F.one_hot(pred10, num_classes=num_classes).sum(dim=1).float() * cats).max(dim=1)[0]
    return hit1.mean(), hit10.mean(), pred1, pred10, hit1.bool()


def get_eval(data_dict, config, answer_vocab=None, use_reference=False, use_lang_classifier=False, use_oracle=False,
             use_cat_rand=False, use_best=False, post_processing=None, host_outputs=True):
    dev = data_dict["objectness_scores"].device
    zero = lambda: torch.zeros((), device=dev)
    objectness_preds_batch = torch.argmax(data_dict["objectness_scores"], 2).long()
    objectness_labels_batch = data_dict["objectness_label"].long()
    if post_processing:
        from .ap_helper import parse_predictions
        _ = parse_predictions(data_dict, post_processing)
        nms_masks = data_dict["pred_mask_device"].long()
        pred_masks = (nms_masks * objectness_preds_batch == 1).float()
    else:
        pred_masks = (objectness_preds_batch == 1).float()
    label_masks = (objectness_labels_batch == 1).float()
    B, K = pred_masks.shape

    cluster_preds = F.one_hot(torch.argmax(data_dict["cluster_ref"] * pred_masks, 1), K).float()
    cluster_labels = data_dict["cluster_labels"].float() * label_masks
    if data_dict["cluster_labels"].dtype == torch.float32:
        data_dict["cluster_labels"] = cluster_labels      # the reference's in-place `*=` (see the module docstring)
    corrects = torch.sum((cluster_preds == 1) * (cluster_labels == 1), dim=1).float()
    ref_acc = corrects / (1.0 + 1e-8)

    if use_best:
        pred_ref = torch.argmax(data_dict["cluster_labels"], 1)
        data_dict["cluster_ref"] = data_dict["cluster_labels"]
    # (a separate `if`, as in the reference: with use_best and without use_cat_rand the `else` below still runs, on the
    # cluster_ref just replaced -- pred_ref = argmax(cluster_labels * pred_masks), cluster_ref = cluster_labels * pred_masks)
    if use_cat_rand:
        sem = data_dict["sem_cls_label"].clone()
        k2 = torch.arange(sem.shape[1], device=dev).unsqueeze(0)
        sem = sem - (k2 >= data_dict["num_bbox"].view(-1, 1)).to(sem.dtype)            # sem_cls_label[num_bbox:] -= 1
        cand = torch.gather(sem == data_dict["object_cat"].view(-1, 1), 1, data_dict["object_assignment"])  # (B, K)
        pick = torch.argmax(torch.rand(B, K, device=dev).clamp_min(1e-6) * cand, 1)
        cluster_preds = F.one_hot(pick, K).float() * cand.any(1, keepdim=True)
        pred_ref = torch.argmax(cluster_preds, 1)
        data_dict["cluster_ref"] = cluster_preds
    else:
        pred_ref = torch.argmax(data_dict["cluster_ref"] * pred_masks, 1)
        data_dict["cluster_ref"] = data_dict["cluster_ref"] * pred_masks

    if use_oracle:
        assign = data_dict["object_assignment"]
        a3 = assign.unsqueeze(2).repeat(1, 1, 3)
        pred_center = torch.gather(data_dict["center_label"], 1, a3)
        pred_heading_class = torch.gather(data_dict["heading_class_label"], 1, assign)
        pred_heading_residual = torch.gather(data_dict["heading_residual_label"], 1, assign).unsqueeze(-1)
        pred_size_class = torch.gather(data_dict["size_class_label"], 1, assign)
        pred_size_residual = torch.gather(data_dict["size_residual_label"], 1, a3)
    else:
        pred_center = data_dict["center"]
        pred_heading_class = torch.argmax(data_dict["heading_scores"], -1)
        pred_heading_residual = torch.gather(data_dict["heading_residuals"], 2, pred_heading_class.unsqueeze(-1)).squeeze(2)
        pred_size_class = torch.argmax(data_dict["size_scores"], -1)
        pred_size_residual = torch.gather(data_dict["size_residuals"], 2,
                                          pred_size_class.unsqueeze(-1).unsqueeze(-1).repeat(1, 1, 1, 3)).squeeze(2)
    data_dict["pred_mask"] = pred_masks
    data_dict["label_mask"] = label_masks
    data_dict["pred_center"] = pred_center
    data_dict["pred_heading_class"] = pred_heading_class
    data_dict["pred_heading_residual"] = pred_heading_residual
    data_dict["pred_size_class"] = pred_size_class
    data_dict["pred_size_residual"] = pred_size_residual

    # ---- the referred box of every sample, prediction and ground truth, decoded in one batch (:171-197) -------------
    gt_ref = torch.argmax(data_dict["ref_box_label"], 1)
    msa = np.asarray(config.mean_size_arr, dtype=np.float64)
    mean_size = _const(("mean_size64", msa.shape, hash(msa.tobytes())), dev, lambda: torch.from_numpy(msa.copy()))
    bi = torch.arange(B, device=dev)

    def obb(center, hcls, hres, scls, sres, idx):
        c = center[bi, idx, 0:3].detach().double()
        hr = hres[bi, idx].detach().double().reshape(B)
        size = mean_size[scls[bi, idx].long()] + sres[bi, idx].detach().double().reshape(B, 3)
        return c, size, _class2angle(config, hcls[bi, idx].long(), hr) * -1

    p_c, p_size, p_head = obb(pred_center, pred_heading_class, pred_heading_residual, pred_size_class, pred_size_residual,
                              pred_ref)
    g_c, g_size, g_head = obb(data_dict["center_label"], data_dict["heading_class_label"],
                              data_dict["heading_residual_label"], data_dict["size_class_label"],
                              data_dict["size_residual_label"], gt_ref)
    pmin, pmax = _aabb_of_obb(p_c, p_size, p_head)        # get_3d_box + get_box3d_min_max (box_util.py:127-144,282-300)
    gmin, gmax = _aabb_of_obb(g_c, g_size, g_head)
    inter = (torch.minimum(pmax, gmax) - torch.maximum(pmin, gmin)).clamp(min=0).prod(-1)
    ious = inter / ((pmax - pmin).prod(-1) + (gmax - gmin).prod(-1) - inter + 1e-8)      # box3d_iou (:113-124)
    pred_bboxes = construct_bbox_corners(p_c, p_size)     # "get_3d_box() will return problematic bboxes" (:193-197)
    gt_bboxes = construct_bbox_corners(g_c, g_size)

    # ---- language / answer accuracies (:200-296) ---------------------------------------------------------------------
    if use_lang_classifier:
        data_dict["lang_acc"] = (torch.argmax(data_dict["lang_scores"], 1) == data_dict["object_cat"]).float().mean()
    else:
        data_dict["lang_acc"] = zero()
    hits = {}
    for suffix in ("", "_scene", "_2d", "_2d3d"):
        key = "answer_scores" + suffix
        if key not in data_dict:
            data_dict["answer_acc_at1" + suffix] = zero()
            data_dict["answer_acc_at10" + suffix] = zero()
            continue
        acc1, acc10, pred1, pred10, hits[suffix] = _answer_acc(data_dict[key], data_dict["answer_cats"])
        data_dict["answer_acc_at1" + suffix] = acc1
        data_dict["answer_acc_at10" + suffix] = acc10
        if suffix != "_2d3d":
            data_dict["pred_answer_scores" + suffix] = data_dict[key]
        if suffix == "":
            data_dict["pred_answers_at10"] = pred10
            data_dict["pred_answers"] = pred1
    if "_2d" in hits and "_scene" in hits:
        data_dict["answer_acc_at1_3d_over_2d"] = (hits["_scene"] & ~hits["_2d"]).float().mean()
        data_dict["answer_acc_at1_2d_over_3d"] = (~hits["_scene"] & hits["_2d"]).float().mean()

    # ---- other statistics (:306-318) ---------------------------------------------------------------------------------
    obj_pred_val = torch.argmax(data_dict["objectness_scores"], 2)
    data_dict["obj_acc"] = torch.sum((obj_pred_val == data_dict["objectness_label"].long()).float()
                                     * data_dict["objectness_mask"]) / (torch.sum(data_dict["objectness_mask"]) + 1e-6)
    sem_cls_label = torch.gather(data_dict["sem_cls_label"], 1, data_dict["object_assignment"])
    sem_match = (sem_cls_label == data_dict["sem_cls_scores"].argmax(-1)).float()
    data_dict["sem_acc"] = (sem_match * pred_masks).sum() / pred_masks.sum()

    rate25, rate50 = (ious >= 0.25).double().mean(), (ious >= 0.5).double().mean()
    if not host_outputs:
        data_dict["ref_acc"], data_dict["ref_iou"] = ref_acc, ious
        data_dict["ref_iou_rate_0.25"], data_dict["ref_iou_rate_0.5"] = rate25, rate50
        data_dict["pred_bboxes"], data_dict["gt_bboxes"] = pred_bboxes, gt_bboxes
        if use_lang_classifier:
            data_dict["pred_langs"] = data_dict["lang_scores"].detach()
        return data_dict
    # ---- the one device -> host hand-over: everything the reference returns as python lists / numpy -------------------
    parts = [ref_acc.double(), ious, rate25.view(1), rate50.view(1), pred_bboxes.reshape(-1), gt_bboxes.reshape(-1)]
    if use_lang_classifier:
        parts.append(data_dict["lang_scores"].detach().double().reshape(-1))
    host = torch.cat(parts).cpu().numpy()
    o = 0

    def take(n):
        nonlocal o
        o += n
        return host[o - n:o]

    data_dict["ref_acc"] = take(B).astype(np.float32).tolist()
    data_dict["ref_iou"] = take(B).tolist()
    data_dict["ref_iou_rate_0.25"], data_dict["ref_iou_rate_0.5"] = float(take(1)[0]), float(take(1)[0])
    data_dict["pred_bboxes"] = list(take(B * 24).reshape(B, 8, 3))
    data_dict["gt_bboxes"] = list(take(B * 24).reshape(B, 8, 3))
    if use_lang_classifier:
        data_dict["pred_langs"] = take(data_dict["lang_scores"].numel()).astype(np.float32).reshape(tuple(data_dict["lang_scores"].shape))
    return data_dict

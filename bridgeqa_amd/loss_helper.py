"""Detection losses of the training step -- mirror of the reference's lib/loss_helper.py:25-193 and the detection part of
get_loss (:355-464): vote loss, objectness loss, box (centre / heading / size) loss and semantic-class loss.  SURVEY.md
§8f rank 1: the caller on the output side of the hot path.  Same function names, data_dict keys and arithmetic; what
changes underneath: nothing is hard-wired to `.cuda()` (the reference's losses cannot run on any other device), the
label / mask tensors are built by comparisons instead of boolean-mask assignment (each of those is a device-to-host
sync through nonzero()), one-hot selections are gathers, and nn_distance broadcasts instead of tiling.

Round 2 widened this file to the rest of get_loss: compute_reference_loss (the reference's numpy
IoU loop, loss_helper.py:196-244, restated on the device: no host round trip, no per-sample Python loop),
compute_lang_classification_loss (:247-251), compute_answer_classification_loss (:254-279), SoftmaxRankingLoss
(lib/loss.py:17-34) and get_loss (:355-464) itself.  All pinned against the reference's own functions executed in the
build container (oracle/gen_golden_loss.py, oracle/gen_golden_qa.py)."""
import numpy as np
import torch
import torch.nn.functional as F

from .nn_distance import huber_loss, nn_distance

FAR_THRESHOLD = 0.6
NEAR_THRESHOLD = 0.3
GT_VOTE_FACTOR = 3  # number of GT votes per point
OBJECTNESS_CLS_WEIGHTS = [0.2, 0.8]  # put larger weights on positive objectness


_CONST = {}


def _const(key, device, make):
    """small constant tensors live on the device once (a host->device copy per step would also break graph capture)"""
    k = (key, str(device))
    if k not in _CONST:
        _CONST[k] = make().to(device)
    return _CONST[k]


def compute_vote_loss(data_dict):
    """min over (predicted vote, GT vote) pairs of the L1 distance, for seeds that lie in an object (loss_helper.py:25-70)"""
    batch_size = data_dict["seed_xyz"].shape[0]
    num_seed = data_dict["seed_xyz"].shape[1]
    vote_xyz = data_dict["vote_xyz"]
    seed_inds = data_dict["seed_inds"].long()
    seed_gt_votes_mask = torch.gather(data_dict["vote_label_mask"], 1, seed_inds)
    seed_inds_expand = seed_inds.view(batch_size, num_seed, 1).repeat(1, 1, 3 * GT_VOTE_FACTOR)
    seed_gt_votes = torch.gather(data_dict["vote_label"], 1, seed_inds_expand)
    seed_gt_votes = seed_gt_votes + data_dict["seed_xyz"].repeat(1, 1, 3)
    vote_xyz_reshape = vote_xyz.view(batch_size * num_seed, -1, 3)
    seed_gt_votes_reshape = seed_gt_votes.view(batch_size * num_seed, GT_VOTE_FACTOR, 3)
    _, _, dist2, _ = nn_distance(vote_xyz_reshape, seed_gt_votes_reshape, l1=True)
    votes_dist, _ = torch.min(dist2, dim=1)
    votes_dist = votes_dist.view(batch_size, num_seed)
    mask = seed_gt_votes_mask.float()
    return torch.sum(votes_dist * mask) / (torch.sum(mask) + 1e-6)


def compute_objectness_loss(data_dict):
    """-> objectness_loss, objectness_label (B,K) int64, objectness_mask (B,K) f32, object_assignment (B,K) int64
    (loss_helper.py:72-115): a proposal is positive within NEAR_THRESHOLD of a GT centre, ignored in the grey zone"""
    aggregated_vote_xyz = data_dict["aggregated_vote_xyz"]
    gt_center = data_dict["center_label"][:, :, 0:3]
    dist1, ind1, _, _ = nn_distance(aggregated_vote_xyz, gt_center)
    euclidean_dist1 = torch.sqrt(dist1 + 1e-6)
    near = euclidean_dist1 < NEAR_THRESHOLD
    objectness_label = near.long()
    objectness_mask = (near | (euclidean_dist1 > FAR_THRESHOLD)).float()
    objectness_scores = data_dict["objectness_scores"]
    weight = _const("objw", objectness_scores.device, lambda: torch.tensor(OBJECTNESS_CLS_WEIGHTS, dtype=torch.float32))
    weight = weight.to(objectness_scores.dtype)
    objectness_loss = F.cross_entropy(objectness_scores.transpose(2, 1), objectness_label, weight=weight, reduction="none")
    objectness_loss = torch.sum(objectness_loss * objectness_mask) / (torch.sum(objectness_mask) + 1e-6)
    return objectness_loss, objectness_label, objectness_mask, ind1


def compute_box_and_sem_cls_loss(data_dict, config):
    """-> center_loss, heading_cls_loss, heading_reg_loss, size_cls_loss, size_reg_loss, sem_cls_loss
    (loss_helper.py:118-193); needs data_dict['object_assignment'] / ['objectness_label'] from the objectness loss"""
    num_heading_bin = config.num_heading_bin
    mean_size_arr = config.mean_size_arr
    object_assignment = data_dict["object_assignment"]

    pred_center = data_dict["center"]
    gt_center = data_dict["center_label"][:, :, 0:3]
    dist1, _, dist2, _ = nn_distance(pred_center, gt_center)
    box_label_mask = data_dict["box_label_mask"]
    objectness_label = data_dict["objectness_label"].float()
    npos = torch.sum(objectness_label) + 1e-6
    centroid_reg_loss1 = torch.sum(dist1 * objectness_label) / npos
    centroid_reg_loss2 = torch.sum(dist2 * box_label_mask) / (torch.sum(box_label_mask) + 1e-6)
    center_loss = centroid_reg_loss1 + centroid_reg_loss2

    heading_class_label = torch.gather(data_dict["heading_class_label"], 1, object_assignment)
    heading_class_loss = F.cross_entropy(data_dict["heading_scores"].transpose(2, 1), heading_class_label, reduction="none")
    heading_class_loss = torch.sum(heading_class_loss * objectness_label) / npos
    heading_residual_label = torch.gather(data_dict["heading_residual_label"], 1, object_assignment)
    heading_residual_normalized_label = heading_residual_label / (np.pi / num_heading_bin)
    # sum(residuals * one_hot(label)) == the residual of the labelled bin
    picked = torch.gather(data_dict["heading_residuals_normalized"], 2, heading_class_label.unsqueeze(-1)).squeeze(-1)
    heading_residual_normalized_loss = huber_loss(picked - heading_residual_normalized_label, delta=1.0)
    heading_residual_normalized_loss = torch.sum(heading_residual_normalized_loss * objectness_label) / npos

    size_class_label = torch.gather(data_dict["size_class_label"], 1, object_assignment)
    size_class_loss = F.cross_entropy(data_dict["size_scores"].transpose(2, 1), size_class_label, reduction="none")
    size_class_loss = torch.sum(size_class_loss * objectness_label) / npos
    size_residual_label = torch.gather(data_dict["size_residual_label"], 1, object_assignment.unsqueeze(-1).repeat(1, 1, 3))
    idx = size_class_label.view(*size_class_label.shape, 1, 1).expand(-1, -1, 1, 3)
    predicted_size_residual_normalized = torch.gather(data_dict["size_residuals_normalized"], 2, idx).squeeze(2)
    msa = np.ascontiguousarray(np.asarray(mean_size_arr, dtype=np.float32))
    mean_sizes = _const(("mean_size", msa.tobytes()), pred_center.device, lambda: torch.from_numpy(msa.copy()))
    mean_size_label = mean_sizes[size_class_label]  # (B,K,3)
    size_residual_label_normalized = size_residual_label / mean_size_label
    size_residual_normalized_loss = torch.mean(
        huber_loss(predicted_size_residual_normalized - size_residual_label_normalized, delta=1.0), -1)
    size_residual_normalized_loss = torch.sum(size_residual_normalized_loss * objectness_label) / npos

    sem_cls_label = torch.gather(data_dict["sem_cls_label"], 1, object_assignment)
    sem_cls_loss = F.cross_entropy(data_dict["sem_cls_scores"].transpose(2, 1), sem_cls_label, reduction="none")
    sem_cls_loss = torch.sum(sem_cls_loss * objectness_label) / npos
    return (center_loss, heading_class_loss, heading_residual_normalized_loss, size_class_loss,
            size_residual_normalized_loss, sem_cls_loss)


# ---- the same three functions as ONE forward launch + ONE backward launch (csrc/detloss.hip) ------------------------------
FUSED_DET_LOSS = [True]   # device tensors only; the torch composition above stays the golden-pinned definition (and the CPU path)
_FUSED_FLOAT = ("seed_xyz", "vote_xyz", "aggregated_vote_xyz", "objectness_scores", "center", "heading_scores",
                "heading_residuals_normalized", "size_scores", "size_residuals_normalized", "sem_cls_scores", "vote_label",
                "center_label", "box_label_mask", "heading_residual_label", "size_residual_label")
_FUSED_INT = ("vote_label_mask", "heading_class_label", "size_class_label", "sem_cls_label")


class _DetLossFn(torch.autograd.Function):
    """(labels, packing, vote_xyz, center, scores...) -> (terms (8,), objectness_label, objectness_mask, object_assignment,
    ratios (2,)).  scores = the six score tensors, or -- packing -- the ONE (B, K, channels) head output they are slices of
    (proposal_module.decode_scores): its gradient then comes back as one tensor instead of six slice gradients that autograd
    would zero-pad and add.  The gradient of every term w.r.t. its input is produced by the forward launch (the loss is a
    few dozen flops per proposal: cheaper to finish than to revisit); the backward scales the buffers by the upstream
    gradient of their term, one launch."""

    @staticmethod
    def forward(ctx, labels, mean_size, packing, vote_xyz, center, *scores):
        from . import _ext
        t = dict(labels)
        t["vote_xyz"], t["center"] = vote_xyz, center
        if packing is None:
            t.update(zip(_ext._DET_SCORES, scores))
        terms, lab, msk, asg, grads = _ext.det_loss_fwd(t, mean_size, NEAR_THRESHOLD, FAR_THRESHOLD, OBJECTNESS_CLS_WEIGHTS[0],
                                                        OBJECTNESS_CLS_WEIGHTS[1], packing=packing)
        ctx.grads, ctx.packing = grads, packing
        ctx.mark_non_differentiable(lab, msk, asg)
        return terms[:8], lab, msk, asg, terms[8:10].detach()

    @staticmethod
    def backward(ctx, g_terms, g_lab, g_msk, g_asg, g_ratio):
        from . import _ext
        outs = _ext.det_loss_bwd(ctx.grads, g_terms.contiguous().float(), packing=ctx.packing)
        tail = (outs["packed"],) if ctx.packing is not None else tuple(outs[n] for n in _ext._DET_SCORES)
        return (None, None, None, outs["vote_xyz"], outs["center"]) + tail


def _fused_ok(data_dict):
    """-> None (torch composition) or (packing or False): device fp32 tensors in the layouts csrc/detloss.hip reads"""
    if not FUSED_DET_LOSS[0]:
        return None
    try:
        from . import _ext
        c = data_dict["center"]
        if not (c.is_cuda and c.dtype == torch.float32):
            return None
        packing = _ext.det_loss_packing(data_dict)
        K = c.shape[1]
        for k in _FUSED_FLOAT:
            v = data_dict[k]
            if not (v.is_cuda and v.dtype == torch.float32):
                return None
            if k in _ext._DET_SCORES:
                if _ext.det_score_ld(v, K) is None:
                    return None
            elif not v.is_contiguous():
                return None
        for k in _FUSED_INT:
            v = data_dict[k]
            if not (v.is_cuda and v.dtype == torch.int64 and v.is_contiguous()):
                return None
        si = data_dict["seed_inds"]
        if not (si.is_cuda and si.dtype in (torch.int32, torch.int64)
                and max(data_dict["heading_scores"].shape[2], data_dict["size_scores"].shape[2],
                        data_dict["sem_cls_scores"].shape[2]) <= 32
                and data_dict["vote_label"].shape[2] == 3 * GT_VOTE_FACTOR
                and data_dict["vote_xyz"].shape[1] % data_dict["seed_xyz"].shape[1] == 0):
            return None
        return packing or False
    except (KeyError, ImportError):
        return None


def _fused_detection_terms(data_dict, config, packing):
    from . import _ext
    msa = np.ascontiguousarray(np.asarray(config.mean_size_arr, dtype=np.float32))
    dev = data_dict["center"].device
    mean_sizes = _const(("mean_size", msa.tobytes()), dev, lambda: torch.from_numpy(msa.copy()))
    diff = ("vote_xyz", "center") + _ext._DET_SCORES
    labels = {k: data_dict[k] for k in _FUSED_FLOAT + _FUSED_INT + ("seed_inds",) if k not in diff}
    labels["seed_inds"] = labels["seed_inds"].contiguous()   # (a column slice of sa1_inds in the model: one small copy)
    if packing:
        # (the slices themselves are still read by the kernel -- through the base's memory)
        labels.update({k: data_dict[k].detach() for k in _ext._DET_SCORES})
        scores = (packing[0],)
    else:
        scores = tuple(data_dict[k] for k in _ext._DET_SCORES)
    return _DetLossFn.apply(labels, mean_sizes, packing or None, data_dict["vote_xyz"], data_dict["center"], *scores)


def get_detection_loss(data_dict, config, loss_weights=None, amplify=10.0):
    """The detection terms of get_loss (loss_helper.py:355-464, `detection=True`): vote + objectness + box + sem-cls with
    the reference's weights dict (default 1.0 each, scripts/train.py passes 1.0 / 0.5 / 1.0 / 0.1) and the x10 at the
    end; fills the same data_dict entries.  Returns (loss, data_dict)."""
    w = loss_weights or {}
    fused = _fused_ok(data_dict)
    if fused is not None:
        # device tensors: the three functions below and their autograd as two launches (csrc/detloss.hip); same entries
        terms, objectness_label, objectness_mask, object_assignment, ratio = _fused_detection_terms(data_dict, config, fused)
        (vote_loss, objectness_loss, center_loss, heading_cls_loss, heading_reg_loss, size_cls_loss, size_reg_loss,
         sem_cls_loss) = terms.unbind(0)
        data_dict["objectness_label"] = objectness_label
        data_dict["objectness_mask"] = objectness_mask
        data_dict["object_assignment"] = object_assignment
        data_dict["pos_ratio"], data_dict["neg_ratio"] = ratio.unbind(0)
    else:
        vote_loss = compute_vote_loss(data_dict)
        objectness_loss, objectness_label, objectness_mask, object_assignment = compute_objectness_loss(data_dict)
        total = float(objectness_label.shape[0] * objectness_label.shape[1])
        data_dict["objectness_label"] = objectness_label
        data_dict["objectness_mask"] = objectness_mask
        data_dict["object_assignment"] = object_assignment
        data_dict["pos_ratio"] = torch.sum(objectness_label.float()) / total
        data_dict["neg_ratio"] = torch.sum(objectness_mask.float()) / total - data_dict["pos_ratio"]
        center_loss, heading_cls_loss, heading_reg_loss, size_cls_loss, size_reg_loss, sem_cls_loss = \
            compute_box_and_sem_cls_loss(data_dict, config)
    box_loss = center_loss + 0.1 * heading_cls_loss + heading_reg_loss + 0.1 * size_cls_loss + size_reg_loss
    data_dict.update(vote_loss=vote_loss, objectness_loss=objectness_loss, center_loss=center_loss,
                     heading_cls_loss=heading_cls_loss, heading_reg_loss=heading_reg_loss, size_cls_loss=size_cls_loss,
                     size_reg_loss=size_reg_loss, sem_cls_loss=sem_cls_loss, box_loss=box_loss)
    loss = (w.get("vote_loss", 1.0) * vote_loss + w.get("objectness_loss", 1.0) * objectness_loss
            + w.get("box_loss", 1.0) * box_loss + w.get("sem_cls_loss", 1.0) * sem_cls_loss)
    loss = loss * amplify
    data_dict["detection_loss"] = loss
    return loss, data_dict


# ---- reference-expression / language / answer losses (loss_helper.py:196-279) and get_loss (:355-464) ------------------
def softmax_ranking_loss(inputs, targets, mask=None):
    """lib/loss.py:17-34 SoftmaxRankingLoss.forward: -sum(log(softmax(x + 1e-8) + 1e-8) * targets) per sample, then the
    mean or the (mask + 1e-8)-weighted mean"""
    assert inputs.shape == targets.shape
    probs = torch.softmax(inputs + 1e-8, dim=1)
    loss = -torch.sum(torch.log(probs + 1e-8) * targets, dim=1)
    if mask is None:
        return loss.mean()
    mask = mask + 1e-8
    return (loss * mask).sum() / mask.sum()


def _class2angle(config, cls, residual):
    """ScannetDatasetConfig.class2angle_batch (VoteNet / ScanRefer data/scannet/model_util_scannet.py -- ABSENT from the
    reference checkout, restated): bin centre + residual, folded into (-pi, pi]"""
    angle = cls.to(residual.dtype) * (2.0 * np.pi / float(config.num_heading_bin)) + residual
    return torch.where(angle > np.pi, angle - 2.0 * np.pi, angle)


def _aabb_of_obb(center, size, heading):
    """axis-aligned min / max corners of get_3d_box_batch(size, heading, center) (utils/box_util.py:302-325): what
    box3d_iou_batch (:146-171) reduces the eight corners to"""
    from .proposal_module import box_corners
    c = box_corners(center, size, heading)
    return c.min(dim=-2).values, c.max(dim=-2).values


def reference_labels(data_dict, config):
    """one-hot (B, K) float: the proposal whose decoded box has the highest IoU with the referred GT box
    (loss_helper.py:205-240), computed on the device in one batch"""
    msa = np.asarray(config.mean_size_arr, dtype=np.float32)
    mean_size = _const(("mean_size", msa.shape, hash(msa.tobytes())), data_dict["center"].device,
                       lambda: torch.from_numpy(msa.copy()))
    center = data_dict["center"].detach().float()                                         # (B,K,3)
    hcls = torch.argmax(data_dict["heading_scores"], -1)                                  # (B,K)
    hres = torch.gather(data_dict["heading_residuals"], 2, hcls.unsqueeze(-1)).squeeze(2).detach().float()
    scls = torch.argmax(data_dict["size_scores"], -1)
    sres = torch.gather(data_dict["size_residuals"], 2,
                        scls.unsqueeze(-1).unsqueeze(-1).repeat(1, 1, 1, 3)).squeeze(2).detach().float()   # (B,K,3)
    pmin, pmax = _aabb_of_obb(center, mean_size[scls] + sres, _class2angle(config, hcls, hres) * -1)
    g_center = data_dict["ref_center_label"].float()[:, 0:3]                              # (B,3)
    g_size = mean_size[data_dict["ref_size_class_label"].long()] + data_dict["ref_size_residual_label"].float()
    g_head = _class2angle(config, data_dict["ref_heading_class_label"].long(), data_dict["ref_heading_residual_label"].float())
    gmin, gmax = _aabb_of_obb(g_center, g_size, g_head * -1)                              # (B,3)
    gmin, gmax = gmin.unsqueeze(1), gmax.unsqueeze(1)
    inter = (torch.minimum(pmax, gmax) - torch.maximum(pmin, gmin)).clamp(min=0).prod(-1)
    vol_p, vol_g = (pmax - pmin).prod(-1), (gmax - gmin).prod(-1)
    iou = inter / (vol_p + vol_g - inter + 1e-8)
    return F.one_hot(iou.argmax(dim=1), iou.shape[1]).to(torch.float32)


def compute_reference_loss(data_dict, config):
    """-> ref_loss, cluster_preds (B,K), cluster_labels (B,K) (loss_helper.py:196-244)"""
    cluster_preds = data_dict["cluster_ref"]
    cluster_labels = reference_labels(data_dict, config)
    loss_ref = softmax_ranking_loss(cluster_preds, cluster_labels.clone(), mask=data_dict["ref_obj_mask"])
    return loss_ref, cluster_preds, cluster_labels


def compute_lang_classification_loss(data_dict):
    loss_lang = F.cross_entropy(data_dict["lang_scores"], data_dict["object_cat"], reduction="none")
    mask = data_dict["ref_obj_mask"] + 1e-8
    return torch.sum(loss_lang * mask) / torch.sum(mask)


def compute_answer_classification_loss(data_dict):
    """the text decoder's LM loss when there is one, else CE over the closed answer set; plus the scene-only and 2D-3D
    branch losses when their scores are present (loss_helper.py:254-279)"""
    if "decoder_loss" in data_dict:
        loss_answer = data_dict["decoder_loss"]
    else:
        loss_answer = F.cross_entropy(data_dict["answer_scores"], data_dict["answer_cat"])
    for key in ("answer_scores_scene", "answer_scores_2d3d"):
        if key in data_dict:
            loss_answer = loss_answer + F.cross_entropy(data_dict[key], data_dict["answer_cat"])
    return loss_answer


def get_loss(data_dict, config, detection=True, use_reference=True, use_lang_classifier=False, use_answer=True,
             loss_weights=None):
    """lib/loss_helper.py:355-464: every term of the training loss with the reference's flags, weights dict (default 1.0
    each) and the final x10; writes the same data_dict entries.  Returns (loss, data_dict)."""
    w = loss_weights or {}
    _, data_dict = get_detection_loss(data_dict, config, loss_weights=w, amplify=1.0)
    dev = data_dict["objectness_label"].device
    zero = lambda: torch.zeros((), device=dev)
    if not detection:
        for k in ("vote_loss", "objectness_loss", "center_loss", "heading_cls_loss", "heading_reg_loss", "size_cls_loss",
                  "size_reg_loss", "sem_cls_loss", "box_loss"):
            data_dict[k] = zero()
    if use_reference:
        ref_loss, _, cluster_labels = compute_reference_loss(data_dict, config)
        data_dict["cluster_labels"], data_dict["ref_loss"] = cluster_labels, ref_loss
    else:
        lab = data_dict["objectness_label"]
        data_dict["cluster_labels"] = lab.new_zeros(lab.shape)
        data_dict["cluster_ref"] = lab.new_zeros(lab.shape).float()
        data_dict["ref_loss"] = zero()
    data_dict["answer_loss"] = compute_answer_classification_loss(data_dict) if use_answer else zero()
    data_dict["lang_loss"] = compute_lang_classification_loss(data_dict) if use_lang_classifier else zero()
    data_dict["align_loss"], data_dict["mae_loss"] = zero(), zero()
    loss = (w.get("vote_loss", 1.0) * data_dict["vote_loss"] + w.get("objectness_loss", 1.0) * data_dict["objectness_loss"]
            + w.get("box_loss", 1.0) * data_dict["box_loss"] + w.get("sem_cls_loss", 1.0) * data_dict["sem_cls_loss"]
            + w.get("ref_loss", 1.0) * data_dict["ref_loss"] + w.get("lang_loss", 1.0) * data_dict["lang_loss"]
            + w.get("answer_loss", 1.0) * data_dict["answer_loss"] + w.get("mae_loss", 1.0) * data_dict["mae_loss"]
            + w.get("align_loss", 1.0) * data_dict["align_loss"])
    loss = loss * 10
    data_dict["loss"] = loss
    return loss, data_dict

"""Post-processing of the detector's proposals -- mirror of the reference's lib/ap_helper.py:40-178 `parse_predictions`
(the evaluation path: lib/eval_helper.py:71, scripts/eval.py:495, scripts/predict.py): decode the
proposal boxes, drop the empty ones, run the (2-D / 3-D / per-class 3-D) greedy NMS, build `pred_mask` and
`batch_pred_map_cls`.  Same function name, arguments (`end_points`, `config_dict` with the reference's keys) and outputs.

What changes underneath: the reference loops over B x K proposals in Python with a device->host sync per scalar, tests
every box against all N points with a scipy Delaunay hull and runs numpy NMS per scene; here the decode is batched on the
device (proposal_module.box_corners), the point-in-box count and the NMS are HIP kernels (csrc/nms.hip), and the host sees
ONE copy of the small result tensors at the end (the list-of-tuples format of batch_pred_map_cls is host data by
definition).  `ScannetDatasetConfig` (absent from the reference checkout): num_heading_bin, num_class and mean_size_arr
are read from config_dict['dataset_config'] (class2angle / class2size restated as in loss_helper._class2angle)."""
import numpy as np
import torch

from . import _ext
from .loss_helper import _class2angle
from .proposal_module import box_corners


def softmax(x):
    """ap_helper.py:33-38 (numpy)"""
    probs = np.exp(x - np.max(x, axis=len(x.shape) - 1, keepdims=True))
    return probs / np.sum(probs, axis=len(x.shape) - 1, keepdims=True)


def parse_predictions(end_points, config_dict):
    cfg = config_dict["dataset_config"]
    center = end_points["center"].detach().float().contiguous()                       # (B, K, 3)
    B, K = center.shape[:2]
    dev = center.device
    hcls = torch.argmax(end_points["heading_scores"], -1)
    hres = torch.gather(end_points["heading_residuals"], 2, hcls.unsqueeze(-1)).squeeze(2).detach().float()
    scls = torch.argmax(end_points["size_scores"], -1)
    sres = torch.gather(end_points["size_residuals"], 2,
                        scls.unsqueeze(-1).unsqueeze(-1).repeat(1, 1, 1, 3)).squeeze(2).detach().float()
    pred_sem_cls = torch.argmax(end_points["sem_cls_scores"], -1)
    sem_probs = torch.softmax(end_points["sem_cls_scores"].detach().float(), -1)
    mean_size = torch.from_numpy(np.asarray(cfg.mean_size_arr, dtype=np.float32)).to(dev)
    heading = _class2angle(cfg, hcls, hres).contiguous()            # class2angle(..., to_label_format=True)
    size = (mean_size[scls] + sres).contiguous()                    # class2size
    corners = box_corners(center, size, heading)                    # get_3d_box, (B, K, 8, 3)
    nonempty = torch.ones(B, K, dtype=torch.bool, device=dev)
    if config_dict["remove_empty_box"]:
        pc = end_points["point_clouds"]
        pc = pc if (pc.dtype == torch.float32 and pc.is_contiguous()) else pc.float().contiguous()
        nonempty = _ext.box_point_count(pc, center, size, heading, cap=5) >= 5       # ap_helper.py:98 `< 5` -> empty
    obj_prob = torch.softmax(end_points["objectness_scores"].detach().float(), -1)[:, :, 1]
    lo, hi = corners.min(dim=2).values, corners.max(dim=2).values   # (B, K, 3)
    if not config_dict["use_3d_nms"]:
        # bird's-eye boxes over the x and z axes (ap_helper.py:108-113)
        zero, one = torch.zeros_like(lo[..., 0]), torch.ones_like(lo[..., 0])
        box = torch.stack([lo[..., 0], lo[..., 2], zero, hi[..., 0], hi[..., 2], one], -1)
        keep = _ext.nms(box, obj_prob, config_dict["nms_iou"], valid=nonempty, old_type=config_dict["use_old_type_nms"])
    else:
        box = torch.cat([lo, hi], -1)
        same = bool(config_dict.get("cls_nms", False))
        keep = _ext.nms(box, obj_prob, config_dict["nms_iou"], cls=pred_sem_cls if same else None, valid=nonempty,
                        old_type=config_dict["use_old_type_nms"], same_cls=same)
    # ---- the one device -> host hand-over
    keep_h, obj_h, cls_h = keep.cpu().numpy(), obj_prob.cpu().numpy(), pred_sem_cls.cpu().numpy()
    corners_h, sem_h = corners.cpu().numpy().astype(np.float64), sem_probs.cpu().numpy()
    assert keep_h.any(axis=1).all()                                  # ap_helper.py:117 `assert(len(pick)>0)`
    end_points["pred_mask"] = keep_h.astype(np.float64)
    end_points["pred_mask_device"] = keep
    batch_pred_map_cls = []
    for i in range(B):
        sel = [j for j in range(K) if keep_h[i, j] and obj_h[i, j] > config_dict["conf_thresh"]]
        if config_dict["per_class_proposal"]:
            cur = []
            for ii in range(cfg.num_class):
                cur += [(ii, corners_h[i, j], sem_h[i, j, ii] * obj_h[i, j]) for j in sel]
            batch_pred_map_cls.append(cur)
        else:
            batch_pred_map_cls.append([(int(cls_h[i, j]), corners_h[i, j], obj_h[i, j]) for j in sel])
    end_points["batch_pred_map_cls"] = batch_pred_map_cls
    return batch_pred_map_cls

"""bridgeqa_amd.eval_helper.get_eval against the reference's own lib/eval_helper.py get_eval (tests/golden/eval.npz, made by
oracle/gen_golden_eval.py).  Variants 0-2 need no kernel (torch ops only) and run on the CPU; variant 3 (post_processing ->
parse_predictions -> the HIP point-count / NMS kernels) is in tests/test_eval_gpu.py."""
import os
import types

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "eval.npz")
SCALARS = ("lang_acc", "answer_acc_at1", "answer_acc_at10", "answer_acc_at1_scene", "answer_acc_at10_scene",
           "answer_acc_at1_2d", "answer_acc_at10_2d", "answer_acc_at1_2d3d", "answer_acc_at10_2d3d",
           "answer_acc_at1_3d_over_2d", "answer_acc_at1_2d_over_3d", "obj_acc", "sem_acc")
TENSORS = ("pred_mask", "label_mask", "cluster_ref", "cluster_labels", "pred_center", "pred_heading_class",
           "pred_heading_residual", "pred_size_class", "pred_size_residual", "pred_answers", "pred_answers_at10")
FLAGS = ("use_reference", "use_lang_classifier", "use_oracle", "use_cat_rand", "use_best")


def load_variant(z, v, device):
    pre = "v%d_in_" % v
    d = {k[len(pre):]: torch.from_numpy(z[k]).to(device) for k in z.files if k.startswith(pre)}
    NH, NS, NC = (int(x) for x in z["v%d_dims" % v])
    cfg = types.SimpleNamespace(num_heading_bin=NH, num_size_cluster=NS, num_class=NC, mean_size_arr=z["v%d_mean_size_arr" % v])
    kw = {k: bool(f) for k, f in zip(FLAGS, z["v%d_flags" % v])}
    if "v%d_post" % v in z.files:
        post = {k: float(x) for k, x in zip(z["post_keys"], z["v%d_post" % v])}
        for k in ("remove_empty_box", "use_3d_nms", "use_old_type_nms", "cls_nms", "per_class_proposal"):
            post[k] = bool(post[k])
        post["dataset_config"] = cfg
        kw["post_processing"] = post
    return d, cfg, kw


def check_variant(z, v, out, host=True):
    g = lambda k: z["v%d_out_%s" % (v, k)]
    for k in SCALARS:
        if "v%d_out_%s" % (v, k) in z.files:
            np.testing.assert_allclose(float(out[k]), float(g(k)), rtol=0, atol=1e-6, err_msg=k)
        else:
            assert k not in out, k
    for k in TENSORS:
        if "v%d_out_%s" % (v, k) not in z.files:
            continue
        a, b = out[k].detach().cpu().numpy(), g(k)
        assert a.shape == b.shape, (k, a.shape, b.shape)
        if np.issubdtype(b.dtype, np.integer):
            assert np.array_equal(a, b), k
        else:
            np.testing.assert_allclose(a, b, rtol=0, atol=1e-6, err_msg=k)
    tolist = (lambda x: x) if host else (lambda x: x.detach().cpu().numpy())
    np.testing.assert_allclose(np.asarray(tolist(out["ref_acc"]), dtype=np.float64), g("ref_acc"), atol=1e-7)
    np.testing.assert_allclose(np.asarray(tolist(out["ref_iou"])), g("ref_iou"), rtol=0, atol=1e-9)
    rates = [float(out["ref_iou_rate_0.25"]), float(out["ref_iou_rate_0.5"])]
    np.testing.assert_allclose(rates, g("ref_iou_rates"), atol=1e-12)
    pb = np.stack([np.asarray(x) for x in out["pred_bboxes"]]) if host else out["pred_bboxes"].cpu().numpy()
    gb = np.stack([np.asarray(x) for x in out["gt_bboxes"]]) if host else out["gt_bboxes"].cpu().numpy()
    np.testing.assert_allclose(pb, g("pred_bboxes"), rtol=0, atol=1e-9)
    np.testing.assert_allclose(gb, g("gt_bboxes"), rtol=0, atol=1e-9)
    if "v%d_out_pred_langs" % v in z.files:
        pl = out["pred_langs"] if host else out["pred_langs"].cpu().numpy()
        assert np.array_equal(np.asarray(pl), g("pred_langs"))
    if host:
        assert isinstance(out["ref_acc"], list) and isinstance(out["ref_iou"], list)
        assert isinstance(out["ref_iou_rate_0.25"], float) and isinstance(out["pred_bboxes"], list)


@pytest.mark.parametrize("v", [0, 1, 2, 4])
@pytest.mark.parametrize("host", [True, False])
def test_get_eval_matches_reference(v, host):
    from bridgeqa_amd.eval_helper import get_eval
    z = np.load(GOLD)
    d, cfg, kw = load_variant(z, v, "cpu")
    out = get_eval(d, cfg, host_outputs=host, **kw)
    check_variant(z, v, out, host)


def test_get_eval_masks_cluster_labels_in_place_like_the_reference():
    from bridgeqa_amd.eval_helper import get_eval
    z = np.load(GOLD)
    d, cfg, kw = load_variant(z, 0, "cpu")
    before = d["cluster_labels"].clone()
    out = get_eval(d, cfg, **kw)
    assert torch.equal(out["cluster_labels"], before * out["label_mask"])


def test_get_eval_cat_rand_picks_a_candidate():
    from bridgeqa_amd.eval_helper import get_eval
    z = np.load(GOLD)
    d, cfg, kw = load_variant(z, 0, "cpu")
    B, K2 = d["sem_cls_label"].shape
    d["num_bbox"] = torch.full((B,), 6, dtype=torch.long)
    sem = d["sem_cls_label"].clone()
    sem[:, 6:] -= 1
    # make sure sample 0 has a candidate and sample 1 has none
    d["object_cat"] = d["object_cat"].clone()
    d["object_cat"][0] = sem[0, d["object_assignment"][0, 0]]
    d["object_cat"][1] = 10 ** 6
    cand = torch.gather(sem == d["object_cat"].view(-1, 1), 1, d["object_assignment"])
    torch.manual_seed(0)
    out = get_eval(d, cfg, use_cat_rand=True, use_lang_classifier=True)
    pick = out["cluster_ref"]
    assert pick.shape == cand.shape and bool(((pick == 0) | (pick == 1)).all())
    assert bool((pick.sum(1) == cand.any(1).float()).all())            # one pick where there is a candidate, none otherwise
    assert bool((pick * (~cand).float() == 0).all())                   # ... and it is a candidate


def test_get_eval_single_sample_and_no_objects_predicted():
    """B = 1, and a batch where no proposal is predicted as an object (pred_mask all zero): the reference then divides
    0 / 0 in sem_acc (eval_helper.py:317) -- NaN there, NaN here -- and the arg-max over an all-zero vector picks proposal 0"""
    from bridgeqa_amd.eval_helper import get_eval
    z = np.load(GOLD)
    d, cfg, kw = load_variant(z, 0, "cpu")
    one = {k: (v[:1].clone() if torch.is_tensor(v) else v) for k, v in d.items()}
    out = get_eval(one, cfg, **kw)
    assert len(out["ref_acc"]) == 1 and len(out["ref_iou"]) == 1 and out["pred_bboxes"][0].shape == (8, 3)
    np.testing.assert_allclose(out["ref_iou"][0], z["v0_out_ref_iou"][0], atol=1e-9)
    d, cfg, kw = load_variant(z, 0, "cpu")
    d["objectness_scores"] = torch.stack([torch.ones_like(d["objectness_scores"][..., 0]),
                                          -torch.ones_like(d["objectness_scores"][..., 1])], -1)
    out = get_eval(d, cfg, **kw)
    assert float(out["pred_mask"].sum()) == 0.0 and bool(torch.isnan(out["sem_acc"]))
    assert out["ref_acc"] == [0.0] * len(out["ref_acc"]) or all(a in (0.0, 1.0) for a in out["ref_acc"])
    assert torch.equal(out["cluster_ref"], torch.zeros_like(out["cluster_ref"]))

"""bridgeqa_amd.blip_itm (BLIP image-text matching + the view ranking of eval_scene_best_views.py) in fp32 against golden
vectors produced by the reference's own models/blip_itm.py (oracle/gen_golden_itm.py): module API, state-dict key set,
arithmetic, ranking and the i2t file format.  CPU only; tests/test_itm_gpu.py drives the HIP path."""
import json
import os

import numpy as np
import torch

from test_fusion_cpu import close, keys_of

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "itm.npz")


def build(dev):
    from bridgeqa_amd import blip_itm
    from bridgeqa_amd.blip_vqa_3d import DEFAULT_BLIP_CONFIG
    from bridgeqa_amd.med import BertConfig
    cfg = BertConfig.from_json_file(DEFAULT_BLIP_CONFIG)
    cfg.num_hidden_layers = 2
    g = np.load(GOLD)
    m = blip_itm.BLIP_ITM(med_config=cfg, image_size=64, vit="base")
    assert keys_of(m, "itm.") == list(g["itm_keys"])
    return g, m.to(dev).eval(), blip_itm


def run_itm(dev, rtol, atol, tmp_path=None):
    g, m, blip_itm = build(dev)
    t = lambda k: torch.from_numpy(g[k]).to(dev)
    text = {"input_ids": t("ids"), "attention_mask": t("am")}
    with torch.no_grad():
        close(m(t("images")[:3], text, match_head="itc"), g["itc"], rtol, atol)
        close(m(t("images")[:3], text, match_head="itm"), g["itm"], rtol, atol)
        feats = blip_itm.encode_views(m, t("images"), batch_size=2)
        names = ["%d.jpg" % i for i in range(feats.shape[0])]
        view, scores = blip_itm.rank_views(m, feats, names, text, question_ids=[7, 8, 9])
    sim = np.asarray(g["sim"])
    for k, q in enumerate(("7", "8", "9")):
        got = np.array([int(n.split(".")[0]) for n in view[q]])
        np.testing.assert_allclose(np.array(scores[q]), g["scores"][k], rtol=rtol, atol=atol)
        if np.abs(np.diff(g["scores"][k])).min() > 4 * atol:   # the ranking itself, when no two scores are within tolerance
            assert np.array_equal(got, np.asarray(g["order"][k]))
        assert sorted(got.tolist()) == list(range(len(got))) and np.all(np.diff(np.array(scores[q])) <= 0)
        np.testing.assert_allclose(sim[k][got], np.array(scores[q]), rtol=rtol, atol=atol)
    if tmp_path is not None:
        path = os.path.join(str(tmp_path), "scene_eval.pkl")
        blip_itm.save_view_map(path, view, scores)
        back = torch.load(path, map_location="cpu")      # utils/blip_utils.py:233-240 load_scene_view_map
        assert set(back) == {"view", "answer", "itm_scores"} and back["view"] == view and back["answer"] == {}
        json.dumps(back["view"])                          # (the reference also accepts a .json of the same dict)


def test_itm_fp32_vs_reference_golden(tmp_path):
    run_itm(torch.device("cpu"), 2e-4, 2e-5, tmp_path)

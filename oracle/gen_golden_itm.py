"""Golden vectors for the BLIP image-text matching model and the view ranking built on it, produced
by the REFERENCE's own models/blip_itm.py BLIP_ITM (:10-70) in this build container, and by the scoring statements of
eval_scene_best_views.py:246-287 run on that model (the script body is not importable; `rank_reference` quotes them).

TEST INFRASTRUCTURE.  Usage:  python oracle/gen_golden_itm.py   -> tests/golden/itm.npz

Shims: as oracle/gen_golden_fusion.py (transformers 5.x compatibility names, timm / fairscale stand-ins, the tokenizer
replaced by an id pass-through -- ids are synthetic).  ViT-B/16 at 64 x 64 pixels (17 tokens), the text encoder with 2
layers of BERT-base width, padded to the script's max_length 70.  Weights are not stored: both sides fill every state-dict
entry from a generator seeded by its name (tests/golden_util.fill_params).
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden", "itm.npz")
sys.path.insert(0, os.path.join(REPO, "oracle"))
sys.path.insert(0, os.path.join(REPO, "tests"))
from gen_golden_fusion import install_shims, keys_of, npy  # noqa: E402
from golden_util import fill_params  # noqa: E402


def make_inputs():
    g = torch.Generator().manual_seed(21)
    n_img, n_q, L = 5, 3, 70
    images = torch.randn(n_img, 3, 64, 64, generator=g)
    ids = torch.zeros(n_q, L, dtype=torch.long)
    am = torch.zeros(n_q, L, dtype=torch.long)
    for i, n in enumerate((9, 14, 6)):
        ids[i, :n] = torch.randint(1000, 28000, (n,), generator=g)
        ids[i, 0], ids[i, n - 1] = 101, 102
        am[i, :n] = 1
    return images, ids, am


def rank_reference(model, images, text):
    """eval_scene_best_views.py:246-287 (image_names = view indices)"""
    image_embeds = model.visual_encoder(images)
    image_feats = F.normalize(model.vision_proj(image_embeds[:, 0, :]), dim=-1)
    text_output = model.text_encoder(text["input_ids"], attention_mask=text["attention_mask"], return_dict=True, mode="text")
    text_feat = F.normalize(model.text_proj(text_output.last_hidden_state[:, 0, :]), dim=-1)
    sim = text_feat @ image_feats.t()
    topk_pred = sim.topk(k=images.shape[0]).indices
    return sim, topk_pred, torch.gather(sim, 1, topk_pred)


def main():
    install_shims()
    import models.blip as rblip

    class Tok(object):
        pad_token_id, sep_token_id, bos_token_id, enc_token_id = 0, 102, 30522, 30523

        def __call__(self, text, **kw):  # the reference calls self.tokenizer(caption, ...): ids pass through
            class Batch(dict):
                __getattr__ = dict.__getitem__

                def to(self, dev):
                    return self
            return Batch(input_ids=text["input_ids"].clone(), attention_mask=text["attention_mask"])
    rblip.init_tokenizer = lambda: Tok()
    import models.blip_itm as ritm
    ritm.init_tokenizer = lambda: Tok()
    cfg_path = os.path.join(REPO, "tests", "golden", "_tmp_med_config.json")
    base = json.load(open(os.path.join(REF, "configs", "med_config.json")))
    base.update(num_hidden_layers=2)
    json.dump(base, open(cfg_path, "w"))
    torch.manual_seed(0)
    model = ritm.BLIP_ITM(med_config=cfg_path, image_size=64, vit="base")
    os.remove(cfg_path)
    out = {"itm_keys": keys_of("itm.", model)}        # fills the parameters from their names
    model.eval()
    images, ids, am = make_inputs()
    text = {"input_ids": ids, "attention_mask": am}
    with torch.no_grad():
        sim, order, scores = rank_reference(model, images, text)
        # BLIP_ITM.forward, both heads, on matched (image, caption) pairs
        itc = model(images[:3], {"input_ids": ids, "attention_mask": am}, match_head="itc")
        itm = model(images[:3], {"input_ids": ids, "attention_mask": am}, match_head="itm")
    out.update(images=images, ids=ids, am=am, sim=sim, order=order, scores=scores, itc=itc, itm=itm)
    np.savez_compressed(OUT, **npy(out))
    print("sim", sim.numpy().round(4).tolist(), "order", order.tolist(), "itm", itm.numpy().round(4).tolist())
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()

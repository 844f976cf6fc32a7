"""Golden vectors for the 2D-3D fusion path, produced by the REFERENCE's own Python
(models/med.py, models/vit.py, models/blip_vqa_3d.py) in this build container.

TEST INFRASTRUCTURE.  Usage:  python oracle/gen_golden_fusion.py   -> tests/golden/fusion_*.npz

The reference depends on third-party packages that are NOT vendored in /root/reference and are
unpinned in its requirements.txt: `transformers` (med.py header names v4.15.0;
this image has 5.15), `timm` (upstream BLIP: 0.4.12), `fairscale`, `icecream`.  Import-time shims,
all for names OUTSIDE the reference tree:
  * icecream.ic                        -> no-op
  * transformers.modeling_utils.{apply_chunking_to_forward, prune_linear_layer,
    find_pruneable_heads_and_indices}  -> re-exported from transformers.pytorch_utils / dummy
  * PreTrainedModel.init_weights       -> self.apply(self._init_weights) (+ tie_weights), v4.15 behaviour
  * PreTrainedModel.get_head_mask      -> [None] * n
  * PreTrainedModel.invert_attention_mask -> (1 - m) * -1e9 (v4.15 fp32 behaviour; 5.x uses finfo.min)
  * timm PatchEmbed / DropPath / trunc_normal_ -> the published timm-0.4.12 algorithms restated with
    torch (Conv2d(3,D,16,16) -> flatten(2) -> transpose(1,2); per-sample Bernoulli keep; nn.init)
  * fairscale checkpoint_wrapper       -> identity
  * transformers.BertTokenizer         -> never called (ids are synthetic); init_tokenizer is patched
Weights are NOT stored: both sides fill every state-dict entry from a generator seeded by the
entry's name (tests/golden_util.fill_params), so equal values <=> equal key sets and shapes.
"""
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from golden_util import fill_params, subsample  # noqa: E402


def install_shims():
    ic = types.ModuleType("icecream")
    ic.ic = lambda *a, **k: None
    sys.modules["icecream"] = ic

    import transformers.modeling_utils as mu
    import transformers.pytorch_utils as pu
    mu.apply_chunking_to_forward = pu.apply_chunking_to_forward
    mu.prune_linear_layer = pu.prune_linear_layer
    mu.find_pruneable_heads_and_indices = lambda *a, **k: (set(), None)

    def init_weights(self):
        self.apply(self._init_weights)
        if hasattr(self, "get_output_embeddings") and self.get_output_embeddings() is not None \
                and getattr(self.config, "tie_word_embeddings", True):
            self.get_output_embeddings().weight = self.get_input_embeddings().weight
    mu.PreTrainedModel.init_weights = init_weights
    mu.PreTrainedModel.post_init = lambda self: None
    mu.PreTrainedModel.get_head_mask = lambda self, head_mask, n, *a, **k: [None] * n

    def invert_attention_mask(self, m):
        ext = m[:, None, :, :] if m.dim() == 3 else m[:, None, None, :]
        return (1.0 - ext.to(torch.float32)) * -1e9
    mu.PreTrainedModel.invert_attention_mask = invert_attention_mask
    mu.PreTrainedModel.get_input_embeddings = lambda self: self.bert.embeddings.word_embeddings \
        if hasattr(self, "bert") else self.embeddings.word_embeddings

    # ---- timm / fairscale stand-ins (published algorithms, not in the reference tree) ----------
    class PatchEmbed(torch.nn.Module):
        def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
            super().__init__()
            self.img_size, self.patch_size = (img_size, img_size), (patch_size, patch_size)
            self.grid_size = (img_size // patch_size, img_size // patch_size)
            self.num_patches = self.grid_size[0] * self.grid_size[1]
            self.proj = torch.nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

        def forward(self, x):
            return self.proj(x).flatten(2).transpose(1, 2)

    class DropPath(torch.nn.Module):
        def __init__(self, drop_prob=None):
            super().__init__()
            self.drop_prob = drop_prob

        def forward(self, x):
            if self.drop_prob == 0.0 or not self.training:
                return x
            keep = 1 - self.drop_prob
            mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
            return x.div(keep) * mask

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__path__ = []
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    mod("timm"); mod("timm.models")
    mod("timm.models.vision_transformer", _cfg=lambda **k: {}, PatchEmbed=PatchEmbed)
    mod("timm.models.registry", register_model=lambda f: f)
    mod("timm.models.layers", trunc_normal_=torch.nn.init.trunc_normal_, DropPath=DropPath)
    mod("timm.models.helpers", named_apply=None, adapt_input_conv=None)
    mod("timm.models.hub", download_cached_file=None)
    mod("fairscale"); mod("fairscale.nn"); mod("fairscale.nn.checkpoint")
    mod("fairscale.nn.checkpoint.checkpoint_activations", checkpoint_wrapper=lambda m, *a, **k: m)

    os.chdir(REF)
    sys.path.insert(0, REF)


class Tok(object):
    pad_token_id, sep_token_id, bos_token_id, enc_token_id = 0, 102, 198, 199


def npy(d):
    return {k: subsample(v.detach().cpu().numpy()) if isinstance(v, torch.Tensor) else np.asarray(v)
            for k, v in d.items()}


def keys_of(prefix, module):
    return np.array(["%s %s" % (k, "x".join(map(str, s))) for k, s in fill_params(module, prefix)])


def main():
    install_shims()
    os.makedirs(OUT, exist_ok=True)
    from models import med as rmed
    from models import vit as rvit
    g = torch.Generator().manual_seed(1)

    # ---------------- ViT (tiny depth, full width so the per-op code paths are the real ones) -----
    torch.manual_seed(0)
    vit = rvit.VisionTransformer(img_size=64, patch_size=16, embed_dim=96, depth=2, num_heads=4, drop_path_rate=0.1)
    out = {"vit_keys": keys_of("visual_encoder.", vit)}
    vit.eval()
    img = torch.randn(2, 3, 64, 64, generator=g)
    out.update(vit_img=img, vit_out=vit(img))
    # bicubic pos-embed resize (vit.py:283-307): 2x2 grid checkpoint -> 4x4 grid model
    ck = torch.randn(1, 1 + 4, 96, generator=g)
    out.update(pos_ckpt=ck, pos_resized=rvit.interpolate_pos_embed(ck, vit))
    np.savez_compressed(os.path.join(OUT, "fusion_vit.npz"), **npy(out))

    # ---------------- twin encoder + LM decoder (small hidden) ------------------------------------
    cfg = rmed.BertConfig(hidden_size=64, num_attention_heads=4, intermediate_size=128, num_hidden_layers=2,
                          vocab_size=200, max_position_embeddings=64, layer_norm_eps=1e-12, hidden_act="gelu",
                          hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, pad_token_id=0)
    cfg.encoder_width = 64
    cfg.add_cross_attention = True
    torch.manual_seed(0)
    twin = rmed.BertModelTwin(config=cfg, add_pooling_layer=False)
    out = {"twin_keys": keys_of("text_encoder.", twin)}
    twin.eval()
    B, L, P, O = 2, 7, 10, 5
    ids = torch.randint(5, 190, (B, L), generator=g)
    am = torch.ones(B, L, dtype=torch.long); am[1, 5:] = 0
    img_e = torch.randn(B, P, 64, generator=g)
    obj_e = torch.randn(B, O, 64, generator=g)
    om = torch.ones(B, O, dtype=torch.long); om[0, 3:] = 0
    r = twin(ids, attention_mask=am, encoder_hidden_states=img_e,
             encoder_attention_mask=torch.ones(B, P, dtype=torch.long), encoder_hidden_states_twin=obj_e,
             encoder_attention_mask_twin=om, return_dict=True, output_attentions=True)
    h2d, h3d = r.last_hidden_state
    out.update(tw_ids=ids, tw_am=am, tw_img=img_e, tw_obj=obj_e, tw_om=om, tw_h2d=h2d, tw_h3d=h3d,
               tw_self2d=r.attentions[-1][0], tw_self3d=r.attentions[-1][1],
               tw_cross2d=r.cross_attentions[-1][0], tw_cross3d=r.cross_attentions[-1][1])
    torch.manual_seed(0)
    dec = rmed.BertLMHeadModel(config=cfg)
    out["dec_keys"] = keys_of("text_decoder.", dec)
    dec.eval()
    La = 5
    aid = torch.randint(5, 190, (B, La), generator=g); aid[:, 0] = 198
    aam = torch.ones(B, La, dtype=torch.long); aam[0, 3:] = 0; aid[0, 3:] = 0
    tgt = aid.masked_fill(aid == 0, -100)
    r = dec(aid, attention_mask=aam, encoder_hidden_states=h2d.detach(), encoder_attention_mask=am, labels=tgt,
            return_dict=True, reduction="none")
    out.update(dec_ids=aid, dec_am=aam, dec_loss=r.loss, dec_logits=r.logits)
    np.savez_compressed(os.path.join(OUT, "fusion_med.npz"), **npy(out))

    # ---------------- whole BLIP_VQA3D (ViT-B/16 width, 2+2 layers, 64x64 image) -------------------
    import models.blip as rblip
    rblip.init_tokenizer = lambda: Tok()
    import models.blip_vqa_3d as rb3
    rb3.init_tokenizer = lambda: Tok()

    class FakeBatch(dict):
        def to(self, dev):
            return self
        __getattr__ = dict.__getitem__

    def fake_tok(self, text, **kw):  # the reference calls self.tokenizer(question, ...): feed ids through
        return FakeBatch(input_ids=text["input_ids"].clone(), attention_mask=text["attention_mask"])
    Tok.__call__ = fake_tok
    cfg_path = os.path.join(OUT, "_tmp_med_config.json")
    import json
    base = json.load(open(os.path.join(REF, "configs", "med_config.json")))
    base.update(num_hidden_layers=2, vocab_size=200, max_position_embeddings=64)
    json.dump(base, open(cfg_path, "w"))
    rb3.DEFAULT_BLIP_CONFIG = cfg_path
    torch.manual_seed(0)
    model = rb3.BLIP_VQA3D(image_size=64, num_answers=10, use_text_decoder=True, share_decoder=True, scene_size=32)
    os.remove(cfg_path)
    out = {"blip_keys": keys_of("blip_model.", model)}
    model.eval()
    img = torch.randn(B, 3, 64, 64, generator=g)
    q = {"input_ids": ids.clone(), "attention_mask": am}
    a = {"input_ids": aid.clone(), "attention_mask": aam}
    obj = torch.randn(B, O, 32, generator=g)
    dd = {}
    loss, fused, qmask = model(img, q, a, scene_object_embeds=obj, scene_object_mask=om, data_dict=dd)
    out.update(bl_img=img, bl_obj=obj, bl_loss=loss, bl_fused=fused, bl_qmask=qmask,
               bl_cross2d=dd["2d_cross_attention"], bl_cross3d=dd["3d_cross_attention"])
    cand = {"input_ids": torch.randint(5, 190, (6, La), generator=g), "attention_mask": torch.ones(6, La, dtype=torch.long)}
    cand["input_ids"][:, 0] = 198
    with torch.no_grad():
        fused_e, scores, _ = model(img, q, cand, train=False, k_test=3, scene_object_embeds=obj,
                                   scene_object_mask=om, data_dict={})
    out.update(bl_cand=cand["input_ids"], bl_rank_all=scores[0], bl_rank_scene=scores[1], bl_rank_2d=scores[2],
               bl_fused_eval=fused_e)
    np.savez_compressed(os.path.join(OUT, "fusion_blip.npz"), **npy(out))
    # ---------------- kernel-shaped fixtures (round 2): head dim 64, widths the HIP kernels take ------------------
    # (attention D = 64, LayerNorm width % 256, GEMM K % 64) so that `-m gpu` tests drive the bf16 HIP path itself
    # against the reference's numbers -- outputs AND gradients.  A fresh generator: the files above stay bit-identical.
    g2 = torch.Generator().manual_seed(7)

    def grads_of(module, names):
        sd = dict(module.named_parameters())
        return {n.replace(".", "_"): sd[n].grad for n in names}

    # gradients of the whole BLIP_VQA3D train-mode forward of fusion_blip.npz (eval-mode modules: no dropout)
    model.zero_grad(set_to_none=True)
    img_l = img.clone().requires_grad_(True)
    obj_l = obj.clone().requires_grad_(True)
    loss, fused, _ = model(img_l, q, a, scene_object_embeds=obj_l, scene_object_mask=om, data_dict={})
    wf = torch.randn(fused.shape, generator=g2)
    (loss.sum() + (fused * wf).sum()).backward()
    out = {"bl_wf": wf, "grad_img": img_l.grad, "grad_obj": obj_l.grad}
    out.update({"grad_" + k: v for k, v in grads_of(model, [
        "visual_encoder.blocks.0.mlp.fc1.weight", "visual_encoder.blocks.1.attn.qkv.bias",
        "visual_encoder.patch_embed.proj.bias", "text_encoder.encoder.layer.0.crossattention.self.value.weight",
        "text_encoder.encoder.layer_twin.1.intermediate.dense.weight", "text_encoder.encoder.layer.1.output.dense.bias",
        "text_decoder.bert.encoder.layer.0.crossattention.self.key.weight", "text_decoder.cls.predictions.bias",
        "linear_scene_object.0.weight"]).items()})
    np.savez_compressed(os.path.join(OUT, "fusion_blip_grad.npz"), **npy(out))

    torch.manual_seed(0)
    vit = rvit.VisionTransformer(img_size=64, patch_size=16, embed_dim=256, depth=2, num_heads=4, drop_path_rate=0.1)
    out = {"vit_keys": keys_of("visual_encoder.", vit)}
    vit.eval()
    img = torch.randn(3, 3, 64, 64, generator=g2)
    wout = torch.randn(3, 17, 256, generator=g2)
    y = vit(img)
    (y * wout).sum().backward()
    out.update(vit_img=img, vit_out=y, vit_wout=wout)
    out.update({"grad_" + k: v for k, v in grads_of(vit, ["patch_embed.proj.weight", "cls_token", "blocks.0.attn.qkv.weight",
                                                          "blocks.0.attn.qkv.bias", "blocks.0.norm1.weight",
                                                          "blocks.1.mlp.fc1.weight", "blocks.1.mlp.fc1.bias",
                                                          "blocks.1.mlp.fc2.weight", "blocks.1.attn.proj.bias",
                                                          "norm.bias"]).items()})
    np.savez_compressed(os.path.join(OUT, "fusion_vit_k.npz"), **npy(out))

    cfgk = rmed.BertConfig(hidden_size=256, num_attention_heads=4, intermediate_size=512, num_hidden_layers=2,
                           vocab_size=200, max_position_embeddings=64, layer_norm_eps=1e-12, hidden_act="gelu",
                           hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, pad_token_id=0)
    cfgk.encoder_width = 256
    cfgk.add_cross_attention = True
    torch.manual_seed(0)
    twin = rmed.BertModelTwin(config=cfgk, add_pooling_layer=False)
    out = {"twin_keys": keys_of("text_encoder.", twin)}
    twin.eval()
    B, L, P, O = 3, 9, 70, 13          # 70 + 9 keys: two 64-key tiles in the 2D cross-attention
    ids = torch.randint(5, 190, (B, L), generator=g2)
    am = torch.ones(B, L, dtype=torch.long); am[1, 6:] = 0
    img_e = torch.randn(B, P, 256, generator=g2).requires_grad_(True)
    obj_e = torch.randn(B, O, 256, generator=g2).requires_grad_(True)
    om = torch.ones(B, O, dtype=torch.long); om[0, 9:] = 0
    r = twin(ids, attention_mask=am, encoder_hidden_states=img_e,
             encoder_attention_mask=torch.ones(B, P, dtype=torch.long), encoder_hidden_states_twin=obj_e,
             encoder_attention_mask_twin=om, return_dict=True, output_attentions=True)
    h2d, h3d = r.last_hidden_state
    w2, w3 = torch.randn(B, L, 256, generator=g2), torch.randn(B, L, 256, generator=g2)
    ((h2d * w2).sum() + (h3d * w3).sum()).backward()
    out.update(tw_ids=ids, tw_am=am, tw_img=img_e, tw_obj=obj_e, tw_om=om, tw_h2d=h2d, tw_h3d=h3d, tw_w2=w2, tw_w3=w3,
               tw_cross2d=r.cross_attentions[-1][0], tw_cross3d=r.cross_attentions[-1][1],
               grad_img=img_e.grad, grad_obj=obj_e.grad)
    out.update({"grad_" + k: v for k, v in grads_of(twin, [
        "encoder.layer.0.attention.self.query.weight", "encoder.layer.0.crossattention.self.key.weight",
        "encoder.layer.0.crossattention.self.value.bias", "encoder.layer_twin.0.crossattention.self.key.weight",
        "encoder.layer.1.intermediate.dense.weight", "encoder.layer.1.intermediate.dense.bias",
        "encoder.layer_twin.1.output.dense.weight", "encoder.layer.0.attention.output.LayerNorm.weight",
        "embeddings.word_embeddings.weight"]).items()})
    torch.manual_seed(0)
    dec = rmed.BertLMHeadModel(config=cfgk)
    out["dec_keys"] = keys_of("text_decoder.", dec)
    dec.eval()
    La = 6
    aid = torch.randint(5, 190, (B, La), generator=g2); aid[:, 0] = 198
    aam = torch.ones(B, La, dtype=torch.long); aam[0, 4:] = 0; aid[0, 4:] = 0
    tgt = aid.masked_fill(aid == 0, -100)
    enc = h2d.detach().clone().requires_grad_(True)
    r = dec(aid, attention_mask=aam, encoder_hidden_states=enc, encoder_attention_mask=am, labels=tgt,
            return_dict=True, reduction="none")
    r.loss.sum().backward()
    out.update(dec_ids=aid, dec_am=aam, dec_loss=r.loss, dec_logits=r.logits, grad_dec_enc=enc.grad)
    out.update({"grad_dec_" + k: v for k, v in grads_of(dec, [
        "bert.embeddings.word_embeddings.weight", "cls.predictions.transform.dense.weight", "cls.predictions.bias",
        "bert.encoder.layer.0.crossattention.self.query.weight", "bert.encoder.layer.1.output.dense.bias"]).items()})
    np.savez_compressed(os.path.join(OUT, "fusion_med_k.npz"), **npy(out))

    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()

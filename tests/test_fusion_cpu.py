"""2D-3D fusion path (ViT / twin MED encoder / LM decoder / BLIP_VQA3D) in fp32 against golden
vectors produced by the reference's own Python (oracle/gen_golden_fusion.py).  Pins the module API,
the state-dict key sets (strict load compatibility) and the arithmetic.  CPU only."""
import numpy as np
import pytest
import torch

from golden_util import fill_params, subsample


def keys_of(module, prefix):
    return ["%s %s" % (k, "x".join(map(str, s))) for k, s in fill_params(module, prefix)]


def close(a, g, rtol=1e-4, atol=1e-5):
    a = subsample(a.detach().float().cpu().numpy())
    np.testing.assert_allclose(a, g, rtol=rtol, atol=atol)


def small_cfg():
    from bridgeqa_amd.med import BertConfig
    return BertConfig(hidden_size=64, num_attention_heads=4, intermediate_size=128, num_hidden_layers=2,
                      vocab_size=200, max_position_embeddings=64, encoder_width=64)


def run_vit(g, dev, rtol, atol):
    from bridgeqa_amd import vit
    m = vit.VisionTransformer(img_size=64, patch_size=16, embed_dim=96, depth=2, num_heads=4, drop_path_rate=0.1)
    assert keys_of(m, "visual_encoder.") == list(g["vit_keys"])
    m = m.to(dev).eval()
    close(m(torch.from_numpy(g["vit_img"]).to(dev)), g["vit_out"], rtol, atol)
    close(vit.interpolate_pos_embed(torch.from_numpy(g["pos_ckpt"]), m.cpu()), g["pos_resized"], 1e-5, 1e-6)


def run_twin_and_decoder(g, dev, rtol, atol):
    from bridgeqa_amd import med
    t = lambda k: torch.from_numpy(g[k]).to(dev)
    twin = med.BertModelTwin(config=small_cfg(), add_pooling_layer=False)
    assert keys_of(twin, "text_encoder.") == list(g["twin_keys"])
    twin = twin.to(dev).eval()
    B, P = g["tw_img"].shape[:2]
    r = twin(t("tw_ids"), attention_mask=t("tw_am"), encoder_hidden_states=t("tw_img"),
             encoder_attention_mask=torch.ones(B, P, dtype=torch.long, device=dev),
             encoder_hidden_states_twin=t("tw_obj"), encoder_attention_mask_twin=t("tw_om"), return_dict=True,
             output_attentions=True)
    h2d, h3d = r.last_hidden_state
    close(h2d, g["tw_h2d"], rtol, atol); close(h3d, g["tw_h3d"], rtol, atol)
    assert len(r.attentions) == 2  # output_attentions=True -> every layer, as the reference
    close(r.attentions[-1][0], g["tw_self2d"], rtol, atol); close(r.attentions[-1][1], g["tw_self3d"], rtol, atol)
    close(r.cross_attentions[-1][0], g["tw_cross2d"], rtol, atol)
    close(r.cross_attentions[-1][1], g["tw_cross3d"], rtol, atol)
    r2 = twin(t("tw_ids"), attention_mask=t("tw_am"), encoder_hidden_states=t("tw_img"),
              encoder_attention_mask=torch.ones(B, P, dtype=torch.long, device=dev),
              encoder_hidden_states_twin=t("tw_obj"), encoder_attention_mask_twin=t("tw_om"),
              output_attentions="last")
    assert len(r2.attentions) == 1  # the non-materialising mode BLIP_VQA3D uses
    close(r2.cross_attentions[-1][1], g["tw_cross3d"], rtol, atol)
    dec = med.BertLMHeadModel(config=small_cfg())
    assert keys_of(dec, "text_decoder.") == list(g["dec_keys"])
    assert dec.cls.predictions.decoder.weight is dec.bert.embeddings.word_embeddings.weight  # tied LM head
    dec = dec.to(dev).eval()
    aid = t("dec_ids")
    r = dec(aid, attention_mask=t("dec_am"), encoder_hidden_states=t("tw_h2d"), encoder_attention_mask=t("tw_am"),
            labels=aid.masked_fill(aid == 0, -100), return_dict=True, reduction="none")
    close(r.loss, g["dec_loss"], rtol, atol * 10)
    close(r.logits, g["dec_logits"], rtol, atol * 10)


def run_blip(g, dev, rtol, atol):
    from bridgeqa_amd.blip_vqa_3d import BLIP_VQA3D, SyntheticTokenizer
    from bridgeqa_amd.med import BertConfig
    cfg = BertConfig(num_hidden_layers=2, vocab_size=200, max_position_embeddings=64)
    m = BLIP_VQA3D(med_config=cfg, image_size=64, num_answers=10, use_text_decoder=True, share_decoder=True,
                   scene_size=32, tokenizer=SyntheticTokenizer(0, 102, 198, 199))
    assert keys_of(m, "blip_model.") == list(g["blip_keys"])  # strict-load compatible with the reference
    m = m.to(dev).eval()
    t = lambda k: torch.from_numpy(g[k]).to(dev)
    gm = dict(np.load(__import__("os").path.join(__import__("conftest").GOLDEN, "fusion_med.npz")))
    tm = lambda k: torch.from_numpy(gm[k]).to(dev)
    q = {"input_ids": tm("tw_ids"), "attention_mask": tm("tw_am")}
    a = {"input_ids": tm("dec_ids"), "attention_mask": tm("dec_am")}
    dd = {}
    loss, fused, qmask = m(t("bl_img"), q, a, scene_object_embeds=t("bl_obj"), scene_object_mask=tm("tw_om"),
                           data_dict=dd)
    close(loss, g["bl_loss"], rtol, atol * 10)
    close(fused, g["bl_fused"], rtol, atol * 10)
    assert torch.equal(qmask.cpu(), torch.from_numpy(g["bl_qmask"]))
    close(dd["2d_cross_attention"], g["bl_cross2d"], rtol, atol)
    close(dd["3d_cross_attention"], g["bl_cross3d"], rtol, atol)
    cand = {"input_ids": t("bl_cand"), "attention_mask": torch.ones_like(t("bl_cand"))}
    with torch.no_grad():
        fused_e, scores, _ = m(t("bl_img"), q, cand, train=False, k_test=3, scene_object_embeds=t("bl_obj"),
                               scene_object_mask=tm("tw_om"), data_dict={})
    close(fused_e, g["bl_fused_eval"], rtol, atol * 10)
    close(scores[1], g["bl_rank_scene"], rtol * 10, atol * 100)
    close(scores[2], g["bl_rank_2d"], rtol * 10, atol * 100)
    close(scores[0], g["bl_rank_all"], rtol * 10, 1e-7)


def test_vit_vs_reference_golden(golden):
    run_vit(golden("fusion_vit.npz"), torch.device("cpu"), 1e-4, 1e-5)


def test_twin_encoder_and_decoder_vs_reference_golden(golden):
    run_twin_and_decoder(golden("fusion_med.npz"), torch.device("cpu"), 1e-4, 1e-5)


def test_blip_vqa3d_vs_reference_golden(golden):
    run_blip(golden("fusion_blip.npz"), torch.device("cpu"), 2e-4, 2e-5)


def test_prepared_token_masks_equal_the_masks_built_in_place(golden):
    """round 6: BLIP_VQA3D.prepare_text also builds the attention masks that depend on the token masks only (question
    self-attention mask and its stacked form, the 2D stream's encoder mask over cat(image tokens, question), the decoder's
    causal mask and its stacked question mask); a forward fed with them (text_prep=) returns exactly what the forward that
    builds every mask in place returns -- loss, fused states, attention maps -- and the same input gradient"""
    from bridgeqa_amd.blip_vqa_3d import BLIP_VQA3D, SyntheticTokenizer
    from bridgeqa_amd.med import BertConfig
    g = golden("fusion_blip.npz")
    gm = dict(np.load(__import__("os").path.join(__import__("conftest").GOLDEN, "fusion_med.npz")))
    cfg = BertConfig(num_hidden_layers=2, vocab_size=200, max_position_embeddings=64)
    torch.manual_seed(3)
    m = BLIP_VQA3D(med_config=cfg, image_size=64, num_answers=10, use_text_decoder=True, share_decoder=True,
                   scene_size=32, tokenizer=SyntheticTokenizer(0, 102, 198, 199)).eval()
    t = lambda k: torch.from_numpy(g[k])
    tm = lambda k: torch.from_numpy(gm[k])
    q = {"input_ids": tm("tw_ids"), "attention_mask": tm("tw_am")}
    a = {"input_ids": tm("dec_ids"), "attention_mask": tm("dec_am")}
    res = []
    for prepared in (False, True):
        obj = t("bl_obj").clone().requires_grad_(True)
        dd = {}
        prep = m.prepare_text(dict(q), dict(a)) if prepared else None
        if prepared:
            assert set(prep["enc_masks"]) == {"ext", "enc_ext"} and set(prep["dec_masks"]) == {"ext", "enc_ext"}
            assert prep["enc_masks"]["enc_ext"].shape[-1] == m.visual_encoder.patch_embed.num_patches + 1 + q["input_ids"].shape[1]
        loss, fused, qmask = m(t("bl_img"), dict(q), dict(a), scene_object_embeds=obj, scene_object_mask=tm("tw_om"),
                               data_dict=dd, text_prep=prep)
        loss.backward()
        res.append((loss.detach(), fused.detach(), dd["2d_cross_attention"], dd["3d_cross_attention"], obj.grad.clone()))
    for x, y in zip(*res):
        assert torch.equal(x, y)


def test_twin_init_copies_2d_stream_and_backward_reaches_both():
    from bridgeqa_amd import med
    torch.manual_seed(0)
    twin = med.BertModelTwin(config=small_cfg(), add_pooling_layer=False)
    twin.init_twin()
    for a, b in zip(twin.encoder.layer.parameters(), twin.encoder.layer_twin.parameters()):
        assert torch.equal(a, b)
    ids = torch.randint(5, 190, (2, 6))
    r = twin(ids, attention_mask=torch.ones(2, 6, dtype=torch.long), encoder_hidden_states=torch.randn(2, 4, 64),
             encoder_attention_mask=torch.ones(2, 4, dtype=torch.long),
             encoder_hidden_states_twin=torch.randn(2, 3, 64),
             encoder_attention_mask_twin=torch.ones(2, 3, dtype=torch.long))
    (r.last_hidden_state[0].sum() + r.last_hidden_state[1].sum()).backward()
    assert twin.encoder.layer[0].crossattention.self.key.weight.grad.abs().sum() > 0
    assert twin.encoder.layer_twin[1].crossattention.self.key.weight.grad.abs().sum() > 0
    assert twin.encoder.layer[0].output.LayerNorms[0].weight.grad is None  # unused extra LN (state-dict parity only)


def test_grad_tap_is_correct_in_either_backward_order():
    """fusion_ops.GradTap: the residual-branch gradient parked by tap() must reach x exactly once, whether the tap fires
    before the consuming linear's backward (the designed order: it then rides on the dX GEMM) or after it (the tap then
    hands its gradient back to autograd)"""
    from bridgeqa_amd import fusion_ops as ops
    torch.manual_seed(0)
    lin = torch.nn.Linear(8, 8)
    x0 = torch.randn(5, 8)

    def run(order):
        x = x0.clone().requires_grad_(True)
        h = x * 1.0                                       # a non-leaf, as a hidden state is
        t = ops.GradTap()
        if order == "designed":                           # consumer first, tap created later -> tap fires first
            y = ops.linear(h, lin.weight, lin.bias, tap=t)
            r = ops.tap(h, t)
        elif order == "reversed":                         # tap created first -> the consumer's backward runs first
            r = ops.tap(h, t)
            y = ops.linear(h, lin.weight, lin.bias, tap=t)
        else:
            y, r = torch.nn.functional.linear(h, lin.weight, lin.bias), h
        ((y * y).sum() + (r * 3.0).sum()).backward()
        return x.grad.clone()

    want = run("plain")
    assert torch.allclose(run("designed"), want, atol=1e-6)
    assert torch.allclose(run("reversed"), want, atol=1e-6)


def test_weight_gradient_launch_plan_saves_the_partial_round():
    """fusion_ops.plan_big_launches: 1296 tiles of equal length on 256 CUs are six rounds as (972 | 324), five when three
    9-tile problems go to the small-tile kernel and the rest is cut (765 | 504)"""
    from bridgeqa_amd.fusion_ops import plan_big_launches
    tiles = [36, 36, 9, 27] * 12                      # fc2, fc1, proj, qkv of 12 ViT blocks (256 x 256 tiles)
    groups, moved = plan_big_launches(tiles, 256)
    seen = sorted([k for g in groups for k in g] + moved)
    assert seen == list(range(len(tiles))) and all(len(g) <= 36 for g in groups)
    rounds = sum(-(-sum(tiles[k] for k in g) // 256) for g in groups)
    assert rounds == 5 and sorted(tiles[k] for k in moved) == [9, 9, 9]
    # nothing to gain: one launch that is already whole rounds / a single short launch
    assert plan_big_launches([32] * 8, 256) == ([list(range(8))], [])
    assert plan_big_launches([18] * 12, 256) == ([list(range(12))], [])
    # more problems than two launches can hold: the plain cut
    many = [4] * 80
    groups, moved = plan_big_launches(many, 256)
    assert moved == [] and [len(g) for g in groups] == [36, 36, 8]

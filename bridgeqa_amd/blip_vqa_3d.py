"""BLIP_VQA3D -- the 2D-3D twin cross-attention fusion model; mirror of the reference's
models/blip_vqa_3d.py:45-597 (ctor keywords, attribute names => state-dict keys, forward
signature and returned tuples, fuse_2d3d, rank_answer, tile, concat_repeat, blip_vqa3d).

Differences, none of which changes a value a caller reads:
  * text goes in either as strings (needs a tokenizer: pass `tokenizer=` or have HF's
    bert-base-uncased available offline) or ALREADY TOKENISED as an object/dict with `input_ids`
    and `attention_mask` -- the synthetic-data path (no vocabulary files in this image);
  * the twin encoder is asked for the LAST layer's attention maps only (the reference
    materialises all 12 and reads maps[-1], blip_vqa_3d.py:275-282);
  * the reference forces a 480-pixel ViT whatever --image_size says; here
    `image_size` is honoured;
  * no ./temp_model.pth round-trip (save_state_dict/reinit_params race).
"""
import os
from argparse import Namespace
from copy import deepcopy

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import fusion_ops as ops
from .med import BertConfig, BertLMHeadModel, BertModelTwin
from .vit import create_vit, interpolate_pos_embed

DEFAULT_BLIP_CONFIG = os.path.join(os.path.dirname(os.path.realpath(__file__)), "configs", "med_config.json")

# ids of the BLIP tokenizer = bert-base-uncased (30522) + [DEC] (bos) + [ENC]  (models/blip.py:319-331)
PAD_TOKEN_ID, SEP_TOKEN_ID, BOS_TOKEN_ID, ENC_TOKEN_ID = 0, 102, 30522, 30523


class SyntheticTokenizer(object):
    """Stand-in carrying only the special-token ids; used when text arrives pre-tokenised."""

    def __init__(self, pad=PAD_TOKEN_ID, sep=SEP_TOKEN_ID, bos=BOS_TOKEN_ID, enc=ENC_TOKEN_ID):
        self.pad_token_id, self.sep_token_id, self.bos_token_id, self.enc_token_id = pad, sep, bos, enc

    def __call__(self, *a, **k):
        raise RuntimeError("no vocabulary available: pass questions/answers as {'input_ids','attention_mask'}")


def init_tokenizer():
    """models/blip.py:319-331; falls back to the id-only stand-in when no vocabulary is reachable."""
    try:
        from transformers import BertTokenizer
        tok = BertTokenizer.from_pretrained("bert-base-uncased", local_files_only=True)
        if len(tok) < 30000:
            raise OSError("bert-base-uncased vocabulary not available")
        tok.add_special_tokens({"bos_token": "[DEC]"})
        tok.add_special_tokens({"additional_special_tokens": ["[ENC]"]})
        tok.enc_token_id = tok.additional_special_tokens_ids[0]
        return tok
    except Exception:
        return SyntheticTokenizer()


def _tokens(text, tokenizer, device, **kw):
    if isinstance(text, dict):
        text = Namespace(**text)
    if hasattr(text, "input_ids"):
        return Namespace(input_ids=text.input_ids.to(device).clone(), attention_mask=text.attention_mask.to(device))
    out = tokenizer(text, return_tensors="pt", **kw).to(device)
    return Namespace(input_ids=out.input_ids, attention_mask=out.attention_mask)


def to_all_answer_score(ans_idx, ans_score, num_answers, batch_size):
    all_answer_score = torch.zeros([batch_size, len(num_answers)], device=ans_idx.device)
    rows = torch.arange(ans_score.size(0), device=ans_idx.device) % batch_size
    all_answer_score.index_put_((rows[:, None].expand_as(ans_idx), ans_idx), ans_score, accumulate=True)
    return torch.where(all_answer_score == 0, -1e6, all_answer_score)


def concat_repeat(a, b, n_repeat):
    assert a.shape == b.shape
    ab = torch.stack((a, b), dim=1).flatten(0, 1)  # [a1, b1, a2, b2, ...]
    return ab.repeat_interleave(n_repeat, dim=0)


def tile(x, dim, n_tile):
    """[x1..xn] -> [x1 * n_tile, x2 * n_tile, ...] along dim (blip_vqa_3d.py:591-597)."""
    return x.repeat_interleave(n_tile, dim=dim)


def _mlp_head(d_in, d_hidden, d_out, pdrop):
    return nn.Sequential(nn.Linear(d_in, d_hidden), nn.GELU(), nn.Dropout(pdrop), nn.LayerNorm(d_hidden),
                         nn.Linear(d_hidden, d_out))


def _adapter(d_in, d):
    return nn.Sequential(nn.Linear(d_in, d), nn.GELU(), nn.Dropout(0.1), nn.Linear(d, d), nn.GELU(), nn.LayerNorm(d))


def _run_adapter(seq, x):
    """the nn.Sequential of _adapter (blip_vqa_3d.py:113-120 linear_scene_object), evaluated through fusion_ops so that on
    the bf16 path its two linears run on the MFMA GEMM kernels with the GELU in the epilogue (as nn.Sequential they were
    fp32 library GEMMs on the critical path); same modules, same state-dict keys"""
    h = ops.linear(x, seq[0].weight, seq[0].bias, act="gelu")
    h = seq[2](h)
    h = ops.linear(h, seq[3].weight, seq[3].bias, act="gelu")
    return ops.layer_norm(h, seq[5])


class BLIP_VQA3D(nn.Module):
    def __init__(self, med_config=DEFAULT_BLIP_CONFIG, image_size=480, vit="base", vit_grad_ckpt=False,
                 vit_ckpt_layer=0, scene_size=128, num_answers=None, use_text_decoder=False, answer_pdrop=0.1,
                 scene_feature_position="paralleltwin", use_scene_weight=False, use_scene_classifier=False,
                 use_scene_classifier_2d3d=False, not_copy_weights=False, mix_tokens=False, share_decoder=False,
                 num_hidden_layers_twin=None, encoder_layers=None, decoder_layers=None, tokenizer=None, **kwargs):
        super().__init__()
        assert num_answers is not None, "num_answers must be specified"
        assert scene_feature_position == "paralleltwin"
        self.num_answers = num_answers
        self.use_text_decoder = use_text_decoder
        self.scene_feature_position = scene_feature_position
        self.use_scene_classifier = use_scene_classifier
        self.use_scene_classifier_2d3d = use_scene_classifier_2d3d
        self.not_copy_weights = not_copy_weights
        self.mix_tokens = mix_tokens
        self.share_decoder = share_decoder

        self.visual_encoder, vision_width = create_vit(vit, image_size, vit_grad_ckpt, vit_ckpt_layer,
                                                       drop_path_rate=0.1)
        self.tokenizer = tokenizer if tokenizer is not None else init_tokenizer()

        def load_cfg():
            return deepcopy(med_config) if isinstance(med_config, BertConfig) else BertConfig.from_json_file(med_config)

        encoder_config = load_cfg()
        if encoder_layers is not None:
            encoder_config.num_hidden_layers = encoder_layers
        encoder_config.encoder_width = vision_width
        if num_hidden_layers_twin is not None:
            encoder_config.num_hidden_layers_twin = num_hidden_layers_twin
        self.text_encoder = BertModelTwin(config=encoder_config, add_pooling_layer=False)

        hidden = encoder_config.hidden_size
        lowrank = hidden // 8
        self.lowrank_2d = nn.Linear(hidden, lowrank)
        self.lowrank_3d = nn.Linear(hidden, lowrank)
        self.bilinear_fusion = nn.Bilinear(lowrank, lowrank, hidden)

        decoder_config = load_cfg()
        if decoder_layers is not None:
            decoder_config.num_hidden_layers = decoder_layers
        self.text_decoder = BertLMHeadModel(config=decoder_config)
        if self.share_decoder:
            self.text_decoder_scene = self.text_decoder
        else:
            self.text_decoder_scene = BertLMHeadModel(config=deepcopy(decoder_config))

        self.answer_cls = _mlp_head(hidden, hidden, num_answers, answer_pdrop)
        self.answer_cls_2d3d = _mlp_head(hidden, hidden, num_answers, answer_pdrop)
        self.linear_scene_object = _adapter(scene_size, decoder_config.hidden_size)
        self.camera_encoder = _adapter(16, decoder_config.hidden_size)
        self.use_scene_weight = use_scene_weight
        self.scene_weight = nn.Parameter(torch.zeros(1, dtype=torch.float32) + 1e-5, requires_grad=True)
        self.pretrained = None
        self.projection_head = nn.Sequential(nn.Linear(vision_width, vision_width), nn.GELU(),
                                             nn.LayerNorm(vision_width), nn.Dropout(0.1),
                                             nn.Linear(vision_width, 1), nn.Sigmoid())
        self.copy_weights()

    @property
    def parallel(self):
        return self.scene_feature_position in ["parallel", "parallel++", "parallelshare", "paralleltwin"]

    def copy_weights(self):
        """3D stream starts as a copy of the 2D stream (blip_vqa_3d.py:183-188)."""
        if self.parallel and not self.not_copy_weights:
            self.text_encoder.init_twin()
            if not self.share_decoder:
                self.text_decoder_scene.load_state_dict(self.text_decoder.state_dict(), strict=False)

    # ------------------------------------------------------------------------------------------
    def prepare_text(self, question, answer=None, device=None):
        """Everything of the train-mode forward that depends on the TOKEN IDS only -- question tokens with [ENC] set and their
        embeddings (word + position -> LayerNorm -> dropout, med.py:81-98), and for the shared answer decoder the answer
        tokens with [BOS], the -100 targets, both stacked for the two streams, and the decoder's embeddings -- so that a
        scheduler can run it BESIDE the image encoder / detector instead of at the head of the fusion chain, and its backward
        (LayerNorm + two embedding tables of 30 524 x 768, one of them tied to the LM head) beside the image backward
        (pipeline.PhasedTrainStep's text_prep / text_prep_bwd phases: ~50 launches of 5 us off the critical chain).
        forward(..., text_prep=<this dict>) consumes it; values identical to computing them in place."""
        dev = device if device is not None else self.text_encoder.embeddings.word_embeddings.weight.device
        q = _tokens(question, self.tokenizer, dev, padding="longest", truncation=True, max_length=80)
        q.input_ids[:, 0] = self.tokenizer.enc_token_id
        prep = {"question": q, "q_embeds": self.text_encoder.embeddings(input_ids=q.input_ids)}
        two = lambda t: torch.cat((t, t), dim=0)
        # the attention masks that depend on the token masks only (round 6: ~40 launches of 5 us that sat at the head of the
        # twin encoder and of the decoder): the question's extended mask, its stacked form for the paired levels, the 2D
        # stream's encoder mask over cat(image tokens, question) -- the image tokens are never masked --, and for the shared
        # decoder the causal answer mask and the stacked question mask its cross-attentions read; converted to the kernels'
        # format here too (fusion_ops.prime_masks).  The 3D stream's mask needs the detector's objectness and stays in _encode.
        te = self.text_encoder
        am = q.attention_mask
        ext = te.get_extended_attention_mask(am, tuple(am.shape), dev, False)
        ext._bq_stacked = two(ext)
        P = self.visual_encoder.patch_embed.num_patches + 1
        enc_ext = te.invert_attention_mask(torch.cat((torch.ones(am.shape[0], P, dtype=am.dtype, device=dev), am), dim=1))
        ops.prime_masks(ext, ext._bq_stacked, enc_ext)
        prep["enc_masks"] = {"ext": ext, "enc_ext": enc_ext}
        if answer is not None and self.use_text_decoder and self.share_decoder and not self.use_scene_classifier:
            a = _tokens(answer, self.tokenizer, dev, padding="longest")
            a.input_ids[:, 0] = self.tokenizer.bos_token_id
            targets = a.input_ids.masked_fill(a.input_ids == self.tokenizer.pad_token_id, -100)
            ids2 = two(a.input_ids)
            att2 = two(a.attention_mask)
            td = self.text_decoder.bert
            dext = td.get_extended_attention_mask(att2, tuple(ids2.shape), dev, True)
            denc = td.invert_attention_mask(two(am))
            ops.prime_masks(getattr(dext, "_bq_causal_key_mask", None), denc)
            prep.update(answer=a, ids2=ids2, att2=att2, targets2=two(targets),
                        a_embeds=td.embeddings(input_ids=ids2), dec_masks={"ext": dext, "enc_ext": denc})
        return prep

    def _encode(self, image, question, image_embeds, scene_object_embeds, scene_object_mask, image_pose,
                image_per_sample, data_dict, text_prep=None):
        if image_per_sample > 1:
            B, P, H = image_embeds.size()
            image_embeds = image_embeds.view(B // P, P * image_per_sample, H)
        dev = image_embeds.device
        # (with prepared masks the all-ones image mask is already inside text_prep["enc_masks"]["enc_ext"]; the encoder
        # falls back to building it when the token count does not match -- several images per sample)
        enc_masks = None if text_prep is None else text_prep.get("enc_masks")
        image_atts = None if enc_masks is not None else torch.ones(image_embeds.size()[:-1], dtype=torch.long, device=dev)
        if text_prep is not None:
            question = text_prep["question"]
        else:
            question = _tokens(question, self.tokenizer, dev, padding="longest", truncation=True, max_length=80)
            question.input_ids[:, 0] = self.tokenizer.enc_token_id
        if self.use_scene_weight:
            scene_object_mask = scene_object_mask * torch.clamp(self.scene_weight, min=0, max=1)
        if scene_object_embeds is not None:
            scene_object_embeds = _run_adapter(self.linear_scene_object, scene_object_embeds)
        out = self.text_encoder(question.input_ids if text_prep is None else None, attention_mask=question.attention_mask,
                                encoder_embeds=None if text_prep is None else text_prep["q_embeds"],
                                encoder_hidden_states=image_embeds, encoder_attention_mask=image_atts,
                                encoder_hidden_states_twin=scene_object_embeds,
                                encoder_attention_mask_twin=scene_object_mask, return_dict=True,
                                output_attentions="last", mask_prep=enc_masks)
        h2d, h3d = out.last_hidden_state
        if data_dict is not None:
            data_dict["2d_self_attention"], data_dict["3d_self_attention"] = out.attentions[-1]
            data_dict["2d_cross_attention"], data_dict["3d_cross_attention"] = out.cross_attentions[-1]
        return Namespace(last_hidden_state=h2d), Namespace(last_hidden_state=h3d), question.attention_mask, \
            image_embeds

    def forward(self, image, question, answer=None, n=None, weights=None, train=True, inference="rank",
                k_test=128, image_embeds=None, scene_object_embeds=None, scene_object_mask=None, image_pose=None,
                image_per_sample=1, embed_image=False, depth_map=None, data_dict=None, text_prep=None):
        """text_prep (extension): the dict prepare_text() returned for the same question / answer"""
        if image_embeds is None:
            image_embeds = self.visual_encoder(image)
        if embed_image:
            return image_embeds, self.projection_head(image_embeds[:, 0].float())
        q2d, q3d, q_mask, image_embeds = self._encode(image, question, image_embeds, scene_object_embeds,
                                                      scene_object_mask, image_pose, image_per_sample, data_dict, text_prep)
        B = image_embeds.size(0)
        dev = image_embeds.device

        if not self.use_text_decoder:  # closed-vocabulary classifier heads (train and eval share the code)
            logits = self.answer_cls(q2d.last_hidden_state[:, 0, :].float())
            answer_score_2d = logits.clone()
            answer_score_scene = self.answer_cls(q3d.last_hidden_state[:, 0, :].float())
            fused = self.fuse_2d3d(q2d, q3d)
            if self.use_scene_classifier_2d3d:
                answer_score_2d3d = self.answer_cls_2d3d(fused[:, 0, :])
                logits = (logits + answer_score_scene + answer_score_2d3d) / 3
            else:
                answer_score_2d3d = None
                logits = (logits + answer_score_scene) / 2
            return (logits, answer_score_2d, answer_score_scene, answer_score_2d3d), fused, q_mask

        if train:
            assert answer is not None, "answer must be specified if use text decoder (free-form answer mode)"
            if text_prep is not None and "a_embeds" in text_prep:
                # (token-only work done ahead of time: stacked ids / masks / targets and the decoder's embeddings)
                dm = text_prep.get("dec_masks")
                out = self.text_decoder(None, attention_mask=text_prep["att2"], encoder_embeds=text_prep["a_embeds"],
                                        encoder_hidden_states=torch.cat((q2d.last_hidden_state, q3d.last_hidden_state), dim=0),
                                        encoder_attention_mask=None if dm is not None else torch.cat((q_mask, q_mask), dim=0),
                                        labels=text_prep["targets2"], return_dict=True, reduction="none", mask_prep=dm)
                return out.loss.sum() / B, self.fuse_2d3d(q2d, q3d), q_mask
            answer = _tokens(answer, self.tokenizer, dev, padding="longest")
            answer.input_ids[:, 0] = self.tokenizer.bos_token_id
            targets = answer.input_ids.masked_fill(answer.input_ids == self.tokenizer.pad_token_id, -100)
            if self.share_decoder and not self.use_scene_classifier:
                # the 2D and 3D answer losses use the SAME decoder (blip_vqa_3d.py:119-122,323-343): one pass over
                # the two streams stacked along the batch gives loss2d + loss3d with half the launches
                two = lambda t: torch.cat((t, t), dim=0)
                out = self.text_decoder(two(answer.input_ids), attention_mask=two(answer.attention_mask),
                                        encoder_hidden_states=torch.cat((q2d.last_hidden_state,
                                                                         q3d.last_hidden_state), dim=0),
                                        encoder_attention_mask=two(q_mask), labels=two(targets), return_dict=True,
                                        reduction="none")
                return out.loss.sum() / B, self.fuse_2d3d(q2d, q3d), q_mask
            out = self.text_decoder(answer.input_ids, attention_mask=answer.attention_mask,
                                    encoder_hidden_states=q2d.last_hidden_state, encoder_attention_mask=q_mask,
                                    labels=targets, return_dict=True, reduction="none")
            loss = out.loss.sum() / B
            if self.use_scene_classifier:
                answer_score_scene = self.answer_cls(q3d.last_hidden_state[:, 0, :].float())
                fused = self.fuse_2d3d(q2d, q3d)
                s2d3d = self.answer_cls_2d3d(fused[:, 0, :]) if self.use_scene_classifier_2d3d else None
                return (loss, answer_score_scene, s2d3d), fused, q_mask
            out3d = self.text_decoder_scene(answer.input_ids, attention_mask=answer.attention_mask,
                                            encoder_hidden_states=q3d.last_hidden_state,
                                            encoder_attention_mask=q_mask, labels=targets, return_dict=True,
                                            reduction="none")
            loss = loss + out3d.loss.sum() / B
            return loss, self.fuse_2d3d(q2d, q3d), q_mask

        if inference == "generate":
            # blip_vqa_3d.py:394-417: ten beams per sample, slots 0-4 attend to the 2D question states, slots 5-9 to the 3D
            # ones (beam search itself: generation.py -- the reference's comes from an unpinned `transformers`, parity
            # unpinned); answers decoded with the tokenizer, or the token ids up to [SEP] when no vocabulary is loaded
            num_beams = 5
            question_states = concat_repeat(q2d.last_hidden_state, q3d.last_hidden_state, num_beams)
            question_atts = torch.repeat_interleave(q_mask, 2 * num_beams, dim=0)
            bos_ids = torch.full((B, 1), self.tokenizer.bos_token_id, dtype=torch.long, device=dev)
            outputs = self.text_decoder.generate(input_ids=bos_ids, max_length=20, min_length=1, num_beams=num_beams * 2,
                                                 eos_token_id=self.tokenizer.sep_token_id,
                                                 pad_token_id=self.tokenizer.pad_token_id,
                                                 encoder_hidden_states=question_states,
                                                 encoder_attention_mask=question_atts)
            if hasattr(self.tokenizer, "decode"):
                answers = [self.tokenizer.decode(o, skip_special_tokens=True) for o in outputs]
            else:
                special = {self.tokenizer.pad_token_id, self.tokenizer.sep_token_id, self.tokenizer.bos_token_id,
                           self.tokenizer.enc_token_id}
                answers = [[t for t in o.tolist() if t not in special] for o in outputs]
            return answers, self.fuse_2d3d(q2d, q3d), q_mask

        # ---- rank answers: one-step shortlist + full re-score (blip_vqa_3d.py:418-500) ----------
        assert answer is not None, "answer must be specified if use text decoder (free-form answer mode)"
        answer = _tokens(answer, self.tokenizer, dev, padding="longest")
        answer.input_ids[:, 0] = self.tokenizer.bos_token_id
        n_ans = answer.input_ids.size(0)

        def scatter_scores(ans_idx, ans_score):
            s = torch.zeros([B, n_ans], device=dev)
            rows = (torch.arange(ans_score.size(0), device=dev) % B)[:, None].expand_as(ans_idx)
            return s.index_put_((rows, ans_idx), ans_score.to(s.dtype), accumulate=True)

        ans_idx, ans_score = self.rank_answer(q2d.last_hidden_state, q_mask, answer.input_ids,
                                              answer.attention_mask, k_test)
        all_answer_score = scatter_scores(ans_idx, ans_score)
        all_answer_score_2d = all_answer_score.clone()
        answer_score_2d3d = None
        if self.use_scene_classifier:
            all_answer_score_scene = torch.softmax(self.answer_cls(q3d.last_hidden_state[:, 0, :].float()), dim=-1)
            all_answer_score = torch.where(all_answer_score == 0, -1e4, all_answer_score)
            if all_answer_score.size(1) < all_answer_score_scene.size(1):
                all_answer_score = F.pad(all_answer_score,
                                         (0, all_answer_score_scene.size(1) - all_answer_score.size(1)),
                                         "constant", -1e4)
            all_answer_score = torch.softmax(all_answer_score, dim=-1)
            if self.use_scene_classifier_2d3d:
                answer_score_2d3d = torch.softmax(self.answer_cls_2d3d(self.fuse_2d3d(q2d, q3d)[:, 0, :]), dim=-1)
                all_answer_score = (all_answer_score + all_answer_score_scene + answer_score_2d3d) / 3
            else:
                all_answer_score = (all_answer_score + all_answer_score_scene) / 2
        else:
            idx3d, score3d = self.rank_answer(q3d.last_hidden_state, q_mask, answer.input_ids,
                                              answer.attention_mask, k_test, use_scene=True)
            all_answer_score_scene = scatter_scores(idx3d, score3d)
            all_answer_score = torch.where(all_answer_score == 0, -1e4, all_answer_score)
            all_answer_score_scene = torch.where(all_answer_score_scene == 0, -1e4, all_answer_score_scene)
            all_answer_score = all_answer_score.exp() + (all_answer_score_scene * 1.05).exp()
        all_answer_score_2d = torch.where(all_answer_score_2d == 0, -1e4, all_answer_score_2d)
        return self.fuse_2d3d(q2d, q3d), \
            (all_answer_score, all_answer_score_scene, all_answer_score_2d, answer_score_2d3d), q_mask

    def fuse_2d3d(self, question_output, question_output_scene):
        """Low-rank bilinear + mean (blip_vqa_3d.py:502-507)."""
        h2d = question_output.last_hidden_state.float()
        h3d = question_output_scene.last_hidden_state.float()
        W = self.bilinear_fusion.weight  # (out, r, r)
        if ops.compute_dtype() == torch.bfloat16 and h2d.is_cuda:
            # the two low-rank projections on the MFMA GEMM family too (they were the step's last fp32 library GEMMs on
            # the critical stream), then one contraction over the r * r outer-product features: out = (a (x) b) . W^T with
            # W viewed (out, r*r) -- a single GEMM (M = B L rows, K = r^2 = 9216) instead of a 94 MB fp32 intermediate
            a = ops.linear(question_output.last_hidden_state, self.lowrank_2d.weight, self.lowrank_2d.bias).float()
            b = ops.linear(question_output_scene.last_hidden_state, self.lowrank_3d.weight, self.lowrank_3d.bias).float()
            x = (a.unsqueeze(-1) * b.unsqueeze(-2)).flatten(-2)                  # (B, L, r * r), row-major (i, j) as W
            out = ops.linear(x, W, self.bilinear_fusion.bias).float()
            return out + (h2d + h3d) / 2.0
        a, b = self.lowrank_2d(h2d), self.lowrank_3d(h3d)  # (B, L, r)
        # nn.Bilinear as two dense contractions: out[.,o] = sum_ij a_i W[o,i,j] b_j + bias_o.  (torch's bilinear
        # runs one small GEMM pair PER OUTPUT FEATURE on the GPU -- 768 x 3 launches per call.)
        t = torch.matmul(a, W.permute(1, 0, 2).reshape(W.shape[1], -1))          # (B, L, out * r)
        out = (t.view(*a.shape[:-1], W.shape[0], W.shape[2]) * b.unsqueeze(-2)).sum(-1)
        if self.bilinear_fusion.bias is not None:
            out = out + self.bilinear_fusion.bias
        return out + (h2d + h3d) / 2.0

    def rank_answer(self, question_states, question_atts, answer_ids, answer_atts, k, use_scene=False):
        """Top-k answers by first-token probability, then exact sequence log-likelihood (:509-566)."""
        text_decoder = self.text_decoder_scene if use_scene else self.text_decoder
        num_ques = question_states.size(0)
        start_ids = answer_ids[0, 0].repeat(num_ques, 1)  # bos token
        start_output = text_decoder(start_ids, encoder_hidden_states=question_states,
                                    encoder_attention_mask=question_atts, return_dict=True, reduction="none")
        logits = start_output.logits[:, 0, :].float()
        answer_first_token = answer_ids[:, 1]
        prob_first_token = F.softmax(logits, dim=1).index_select(dim=1, index=answer_first_token)
        k = min(prob_first_token.size(1), k)
        _, topk_ids = prob_first_token.topk(k, dim=1)
        input_ids = answer_ids[topk_ids.reshape(-1)]   # (num_ques*k, La): question-major, as the reference's loop
        input_atts = answer_atts[topk_ids.reshape(-1)]
        targets_ids = input_ids.masked_fill(input_ids == self.tokenizer.pad_token_id, -100)
        question_states = tile(question_states, 0, k)
        question_atts = tile(question_atts, 0, k)
        output = text_decoder(input_ids, attention_mask=input_atts, encoder_hidden_states=question_states,
                              encoder_attention_mask=question_atts, labels=targets_ids, return_dict=True,
                              reduction="none")
        log_probs_sum = (-output.loss).view(num_ques, k)
        return topk_ids, log_probs_sum


def load_checkpoint(model, filename):
    """models/blip.py:371-399 for local files: bicubic pos-embed resize, drop shape-mismatched keys."""
    if not os.path.isfile(filename):
        raise RuntimeError("checkpoint url or path is invalid")
    state_dict = torch.load(filename, map_location="cpu")["model"]
    state_dict["visual_encoder.pos_embed"] = interpolate_pos_embed(state_dict["visual_encoder.pos_embed"],
                                                                   model.visual_encoder)
    own = model.state_dict()
    for key in list(state_dict.keys()):
        if key in own and state_dict[key].shape != own[key].shape:
            del state_dict[key]
    msg = model.load_state_dict(state_dict, strict=False)
    return model, msg


def blip_vqa3d(pretrained="", random_init_blip=False, **kwargs):
    model = BLIP_VQA3D(**kwargs)
    if pretrained:
        model, _ = load_checkpoint(model, pretrained)
        model.copy_weights()
        model.pretrained = pretrained
    if random_init_blip:
        def weight_reset(m):
            f = getattr(m, "reset_parameters", None)
            if callable(f):
                f()
        model.text_encoder.apply(weight_reset)
        model.text_decoder.apply(weight_reset)
        model.text_decoder_scene.apply(weight_reset)
    return model

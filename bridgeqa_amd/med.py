"""MED ("mixture of encoder-decoder") BERT with the 2D/3D twin encoder -- mirror of the live
classes of the reference's models/med.py (BertEmbeddings :53-95, BertSelfAttention :98-226,
BertSelfOutput :229-240, BertAttention :243-289, BertIntermediate :292-304, BertOutput(Parallel)
:307-330, BertLayer :332-397, BertEncoder :400-506, BertEncoderTwin :508-645, prediction heads
:663-707, BertModel :733-973, BertModelTwin :975-1156, BertLMHeadModel :1324-1470).

The reference builds these on HuggingFace `transformers` (v4.15-era PreTrainedModel, ACT2FN,
apply_chunking_to_forward, invert_attention_mask ...), which is not vendored; nothing here
imports transformers.  What is kept exactly: class and attribute names (=> state-dict keys,
including the persistent `embeddings.position_ids` buffer and the unused
`output.LayerNorms.0.*` of every BertOutputParallel), forward keyword names, returned fields,
initialisation (N(0, 0.02), LayerNorm 1/0, zero biases), mask conventions.

MI355X notes: every dense op goes through fusion_ops (bf16 MFMA GEMMs, fp32 softmax / LN);
attention probabilities are only materialised for layers whose maps are actually returned --
the reference calls the twin encoder with output_attentions=True but reads only the LAST layer's
maps (blip_vqa_3d.py:281-282), so `output_attentions="last"` is what BLIP_VQA3D passes.
"""
import json
import math
import os
from types import SimpleNamespace

import torch
import torch.nn as nn

from . import fusion_ops as ops


class BertConfig(object):
    """The fields of configs/med_config.json:1-21 plus the HF defaults med.py relies on."""

    def __init__(self, **kw):
        self.vocab_size = 30524
        self.hidden_size = 768
        self.num_hidden_layers = 12
        self.num_attention_heads = 12
        self.intermediate_size = 3072
        self.hidden_act = "gelu"
        self.hidden_dropout_prob = 0.1
        self.attention_probs_dropout_prob = 0.1
        self.max_position_embeddings = 512
        self.type_vocab_size = 2
        self.initializer_range = 0.02
        self.layer_norm_eps = 1e-12
        self.pad_token_id = 0
        self.encoder_width = 768
        self.add_cross_attention = True
        self.position_embedding_type = "absolute"
        self.chunk_size_feed_forward = 0
        self.output_attentions = False
        self.output_hidden_states = False
        self.use_return_dict = True
        self.use_cache = True
        self.is_decoder = False
        for k, v in kw.items():
            setattr(self, k, v)

    @classmethod
    def from_json_file(cls, path):
        with open(path) as f:
            return cls(**json.load(f))

    def to_dict(self):
        return dict(self.__dict__)


class ModelOutput(SimpleNamespace):
    """Attribute + integer access (non-None fields in declaration order), like HF's ModelOutput."""

    def __getitem__(self, i):
        if isinstance(i, str):
            return getattr(self, i)
        return tuple(v for v in self.__dict__.values() if v is not None)[i]


class BertEmbeddings(nn.Module):
    """word + absolute position embeddings -> LayerNorm -> dropout (no token-type embeddings)."""

    def __init__(self, config):
        super().__init__()
        self.word_embeddings = nn.Embedding(config.vocab_size, config.hidden_size, padding_idx=config.pad_token_id)
        self.position_embeddings = nn.Embedding(config.max_position_embeddings, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        self.register_buffer("position_ids", torch.arange(config.max_position_embeddings).expand((1, -1)))
        self.position_embedding_type = getattr(config, "position_embedding_type", "absolute")
        self.config = config

    def forward(self, input_ids=None, position_ids=None, inputs_embeds=None, past_key_values_length=0):
        seq_length = input_ids.size(1) if input_ids is not None else inputs_embeds.size(1)
        if position_ids is None:
            position_ids = self.position_ids[:, past_key_values_length:seq_length + past_key_values_length]
        if inputs_embeds is None:
            inputs_embeds = self.word_embeddings(input_ids)
        embeddings = inputs_embeds + self.position_embeddings(position_ids)
        return self.dropout(ops.layer_norm(embeddings, self.LayerNorm))


class HoistedStates(object):
    """encoder_hidden_states whose K/V projection for this layer is slot `slot` of `hoisted` (ops.HoistedKV): the
    decoder's 12 cross-attentions read the same question states, so their key / value linears run as one GEMM."""

    def __init__(self, hoisted, slot):
        self.hoisted, self.slot = hoisted, slot


class BertSelfAttention(nn.Module):
    def __init__(self, config, is_cross_attention):
        super().__init__()
        self.config = config
        if config.hidden_size % config.num_attention_heads != 0:
            raise ValueError("The hidden size (%d) is not a multiple of the number of attention heads (%d)"
                             % (config.hidden_size, config.num_attention_heads))
        if getattr(config, "position_embedding_type", "absolute") != "absolute":
            raise NotImplementedError("relative position embeddings are not used by BridgeQA")
        self.num_attention_heads = config.num_attention_heads
        self.attention_head_size = config.hidden_size // config.num_attention_heads
        self.all_head_size = self.num_attention_heads * self.attention_head_size
        self.query = nn.Linear(config.hidden_size, self.all_head_size)
        kv_in = config.encoder_width if is_cross_attention else config.hidden_size
        self.key = nn.Linear(kv_in, self.all_head_size)
        self.value = nn.Linear(kv_in, self.all_head_size)
        self.dropout = nn.Dropout(config.attention_probs_dropout_prob)
        self.save_attention = False

    def save_attn_gradients(self, attn_gradients):
        self.attn_gradients = attn_gradients

    def get_attn_gradients(self):
        return self.attn_gradients

    def save_attention_map(self, attention_map):
        self.attention_map = attention_map

    def get_attention_map(self):
        return self.attention_map

    def _heads(self, x):
        return x.view(x.shape[0], x.shape[1], self.num_attention_heads, self.attention_head_size)

    def forward(self, hidden_states, attention_mask=None, head_mask=None, encoder_hidden_states=None,
                encoder_attention_mask=None, past_key_value=None, output_attentions=False, tap=None):
        """tap: an ops.GradTap -- the projection that reads hidden_states adds the gradient parked there (the residual
        branch of the following BertSelfOutput) to its input gradient"""
        if head_mask is not None:
            raise NotImplementedError("head_mask is always None on the BridgeQA path")
        is_cross = encoder_hidden_states is not None
        want = output_attentions or (is_cross and self.save_attention)
        H, D = self.num_attention_heads, self.attention_head_size
        p_drop = self.dropout.p if self.training else 0.0
        hoisted_kv = isinstance(encoder_hidden_states, HoistedStates)
        # the fused kernels also serve output_attentions (the map is rebuilt from the LSE, detached); a caller that
        # differentiates through the map -- save_attention + the attn_gradients hook -- gets the reference composition
        hooked = is_cross and self.save_attention
        if (not hooked and past_key_value is None
                and ops.compute_dtype() == torch.bfloat16 and (hidden_states.is_cuda or hoisted_kv)):
            # fused projections: Q/K/V (self) or K/V (cross) as ONE GEMM over the shared input, and the attention
            # kernels read / write the packed tensors in place
            B, L = hidden_states.shape[:2]
            rp = bool(output_attentions)
            probs = None
            if is_cross and isinstance(encoder_hidden_states, HoistedStates):
                es = encoder_hidden_states
                q = self._heads(ops.linear(hidden_states, self.query.weight, self.query.bias, tap=tap))
                kv = es.hoisted.kv(es.slot)
                ctx = ops.attention_q_kv(q, kv, 1.0 / math.sqrt(D), p_drop, encoder_attention_mask, return_probs=rp,
                                         sink=(es.hoisted, es.slot))
                present = (kv[:, :, 0].permute(0, 2, 1, 3), kv[:, :, 1].permute(0, 2, 1, 3))
            elif is_cross:
                Lk = encoder_hidden_states.shape[1]
                q = self._heads(ops.linear(hidden_states, self.query.weight, self.query.bias, tap=tap))
                kv = ops.multi_linear(encoder_hidden_states, (self.key, self.value)).view(B, Lk, 2, H, D)
                ctx = ops.attention_q_kv(q, kv, 1.0 / math.sqrt(D), p_drop, encoder_attention_mask, return_probs=rp)
                present = (kv[:, :, 0].permute(0, 2, 1, 3), kv[:, :, 1].permute(0, 2, 1, 3))
            else:
                qkv = ops.multi_linear(hidden_states, (self.query, self.key, self.value), tap=tap).view(B, L, 3, H, D)
                key_mask = getattr(attention_mask, "_bq_causal_key_mask", None)
                if key_mask is not None and ops.packed_kernel_ok(qkv, key_mask):
                    # decoder: the (B,1,L,L) mask is causal AND key padding -- the kernels take it factored
                    ctx = ops.attention_packed(qkv, 1.0 / math.sqrt(D), p_drop, key_mask, causal=True, return_probs=rp)
                elif attention_mask is not None and not (attention_mask.shape[1] == 1 and attention_mask.shape[2] == 1):
                    ctx = ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], attention_mask,
                                        1.0 / math.sqrt(D), return_probs=rp, dropout_p=p_drop)  # causal decoder mask
                    ctx = ctx if rp else ctx[0]
                else:
                    ctx = ops.attention_packed(qkv, 1.0 / math.sqrt(D), p_drop, attention_mask, return_probs=rp)
                present = (qkv[:, :, 1].permute(0, 2, 1, 3), qkv[:, :, 2].permute(0, 2, 1, 3))
            if rp:
                ctx, probs = ctx
                return (ctx.reshape(B, L, self.all_head_size), probs, present)
            return (ctx.reshape(B, L, self.all_head_size), present)
        if hoisted_kv:
            raise RuntimeError("factored encoder states (hoisted K/V projections) reached the reference composition: "
                               "they need the kernel path (bf16 compute, no past_key_value, no save_attention hook)")
        q = self._heads(ops.linear(hidden_states, self.query.weight, self.query.bias, tap=tap))
        src = encoder_hidden_states if is_cross else hidden_states
        k = self._heads(ops.linear(src, self.key.weight, self.key.bias))
        v = self._heads(ops.linear(src, self.value.weight, self.value.bias))
        if is_cross:
            attention_mask = encoder_attention_mask
        elif past_key_value is not None:  # (B,H,Lpast,D) as in the reference's cache layout
            k = torch.cat([past_key_value[0].permute(0, 2, 1, 3), k], dim=1)
            v = torch.cat([past_key_value[1].permute(0, 2, 1, 3), v], dim=1)
        present = (k.permute(0, 2, 1, 3), v.permute(0, 2, 1, 3))
        ctx, probs = ops.attention(q, k, v, attention_mask, 1.0 / math.sqrt(self.attention_head_size),
                                   return_probs=want, dropout_p=self.dropout.p if self.training else 0.0)
        if is_cross and self.save_attention:
            self.save_attention_map(probs)
            if probs.requires_grad:
                probs.register_hook(self.save_attn_gradients)
        ctx = ctx.reshape(ctx.shape[0], ctx.shape[1], self.all_head_size)
        outputs = (ctx, probs) if output_attentions else (ctx,)
        return outputs + (present,)


class BertSelfOutput(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, hidden_states, input_tensor):
        h = ops.linear(hidden_states, self.dense.weight, self.dense.bias)
        return ops.dropout_add_layer_norm(h, input_tensor, self.LayerNorm, self.dropout.p, self.training)


class BertAttention(nn.Module):
    def __init__(self, config, is_cross_attention=False):
        super().__init__()
        self.self = BertSelfAttention(config, is_cross_attention)
        self.output = BertSelfOutput(config)
        self.pruned_heads = set()

    def forward(self, hidden_states, attention_mask=None, head_mask=None, encoder_hidden_states=None,
                encoder_attention_mask=None, past_key_value=None, output_attentions=False):
        t = ops.GradTap()  # the residual gradient of hidden_states rides on the query / QKV projection's dX GEMM
        self_outputs = self.self(hidden_states, attention_mask, head_mask, encoder_hidden_states,
                                 encoder_attention_mask, past_key_value, output_attentions, tap=t)
        attention_output = self.output(self_outputs[0], ops.tap(hidden_states, t))
        return (attention_output,) + self_outputs[1:]


class BertIntermediate(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.intermediate_size)
        if config.hidden_act != "gelu":
            raise NotImplementedError("med_config.json uses hidden_act=gelu")
        self.intermediate_act_fn = ops.gelu

    def forward(self, hidden_states):
        return ops.linear(hidden_states, self.dense.weight, self.dense.bias, act="gelu")


class BertOutput(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.intermediate_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, hidden_states, input_tensor):
        h = ops.linear(hidden_states, self.dense.weight, self.dense.bias)
        return ops.dropout_add_layer_norm(h, input_tensor, self.LayerNorm, self.dropout.p, self.training)


class BertOutputParallel(BertOutput):
    """BertOutput with one extra, selectable LayerNorm (layernorm_idx is 0 on every live path, so
    LayerNorms[0] is a parameter that never receives a gradient -- kept for state-dict parity)."""

    def __init__(self, config):
        super().__init__(config)
        self.LayerNorms = nn.ModuleList([nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
                                         for _ in range(1)])

    def forward(self, hidden_states, input_tensor, layernorm_idx=0, intermediate=None):
        """intermediate: the BertIntermediate whose dense + GELU feeds this layer; given, `hidden_states` is ITS input
        and both linears run as one fused autograd node (ops.mlp)"""
        if intermediate is not None:
            if input_tensor is hidden_states:  # the block's input is also its residual: tap its gradient into the MLP's dX
                t = ops.GradTap()
                h = ops.mlp(hidden_states, intermediate.dense, self.dense, tap=t)
                input_tensor = ops.tap(input_tensor, t)
            else:
                h = ops.mlp(hidden_states, intermediate.dense, self.dense)
        else:
            h = ops.linear(hidden_states, self.dense.weight, self.dense.bias)
        ln = self.LayerNorm if layernorm_idx == 0 else self.LayerNorms[layernorm_idx - 1]
        return ops.dropout_add_layer_norm(h, input_tensor, ln, self.dropout.p, self.training)


class BertLayer(nn.Module):
    """self-attention -> (mode 'multimodal') cross-attention -> FFN, all post-LN."""

    def __init__(self, config, layer_num):
        super().__init__()
        self.config = config
        self.chunk_size_feed_forward = config.chunk_size_feed_forward
        self.seq_len_dim = 1
        self.attention = BertAttention(config)
        self.layer_num = layer_num
        if self.config.add_cross_attention:
            self.crossattention = BertAttention(config, is_cross_attention=self.config.add_cross_attention)
        self.intermediate = BertIntermediate(config)
        self.output = BertOutputParallel(config)

    def forward(self, hidden_states, attention_mask=None, head_mask=None, encoder_hidden_states=None,
                encoder_attention_mask=None, past_key_value=None, output_attentions=False, mode=None,
                layernorm_idx=0):
        self_past = past_key_value[:2] if past_key_value is not None else None
        self_out = self.attention(hidden_states, attention_mask, head_mask, output_attentions=output_attentions,
                                  past_key_value=self_past)
        attention_output = self_out[0]
        outputs = self_out[1:-1]
        present_key_value = self_out[-1]
        if mode == "multimodal":
            assert encoder_hidden_states is not None, "encoder_hidden_states must be given for cross-attention layers"
            cross_out = self.crossattention(attention_output, attention_mask, head_mask, encoder_hidden_states,
                                            encoder_attention_mask, output_attentions=output_attentions)
            attention_output = cross_out[0]
            outputs = outputs + cross_out[1:-1]
        layer_output = self.feed_forward_chunk(attention_output, layernorm_idx)
        return (layer_output,) + outputs + (present_key_value,)

    def feed_forward_chunk(self, attention_output, layernorm_idx):
        # (BertIntermediate -> BertOutputParallel, med.py:292-330; fused: see BertOutputParallel.forward)
        return self.output(attention_output, attention_output, layernorm_idx=layernorm_idx,
                           intermediate=self.intermediate)


_TWIN_BATCH = [True]  # both text streams of a twin level as one stacked batch
_HOIST_CROSS_KV = True  # plain encoder / decoder: all layers' cross K/V in one GEMM


def _wants(output_attentions, i, last):
    return output_attentions is True or (output_attentions == "last" and i == last)


class BertEncoder(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.layer = nn.ModuleList([BertLayer(config, i) for i in range(config.num_hidden_layers)])
        self.gradient_checkpointing = False

    def forward(self, hidden_states, attention_mask=None, head_mask=None, encoder_hidden_states=None,
                encoder_attention_mask=None, past_key_values=None, use_cache=None, output_attentions=False,
                output_hidden_states=False, return_dict=True, mode="multimodal", layernorm_idx=0,
                forward_layers=None):
        all_hidden_states = () if output_hidden_states else None
        all_self_attentions = () if output_attentions else None
        all_cross_attentions = () if output_attentions and self.config.add_cross_attention else None
        next_decoder_cache = () if use_cache else None
        layers = [i for i in range(self.config.num_hidden_layers) if forward_layers is None or i in forward_layers]
        hoisted = None
        key_only = lambda m: m is None or (m.dim() == 4 and m.shape[1] == 1 and m.shape[2] == 1)
        if (_HOIST_CROSS_KV and mode == "multimodal" and len(layers) > 1 and past_key_values is None
                and torch.is_tensor(encoder_hidden_states) and encoder_hidden_states.is_cuda
                and ops.compute_dtype() == torch.bfloat16 and key_only(encoder_attention_mask)
                and self.layer[0].crossattention.self.attention_head_size == 64
                and not any(self.layer[i].crossattention.self.save_attention for i in layers)
                and all(lin.bias is not None and ops._param_ok(lin.weight, lin.bias) for i in layers
                        for lin in (self.layer[i].crossattention.self.key, self.layer[i].crossattention.self.value))):
            # every layer's cross-attention reads the SAME states: their key / value projections as one GEMM (and one dX,
            # one weight-gradient record) instead of one small launch per layer
            hoisted = ops.HoistedKV(ops._c(encoder_hidden_states), [self.layer[i].crossattention.self for i in layers],
                                    self.config.num_attention_heads)
        for n, i in enumerate(layers):
            if output_hidden_states:
                all_hidden_states = all_hidden_states + (hidden_states,)
            past_key_value = past_key_values[i] if past_key_values is not None else None
            want = _wants(output_attentions, i, layers[-1])
            layer_outputs = self.layer[i](hidden_states, attention_mask, None,
                                          HoistedStates(hoisted, n) if hoisted is not None else encoder_hidden_states,
                                          encoder_attention_mask, past_key_value, want, mode=mode,
                                          layernorm_idx=layernorm_idx)
            hidden_states = layer_outputs[0]
            if use_cache:
                next_decoder_cache += (layer_outputs[-1],)
            if want:
                all_self_attentions = all_self_attentions + (layer_outputs[1],)
                if all_cross_attentions is not None and mode == "multimodal":
                    all_cross_attentions = all_cross_attentions + (layer_outputs[2],)
        if output_hidden_states:
            all_hidden_states = all_hidden_states + (hidden_states,)
        if hoisted is not None:
            # (HoistedKV -> its outputs -> their autograd node -> ctx.hold -> HoistedKV is a reference cycle through C++
            # autograd nodes that Python's collector cannot see: the whole autograd graph of a forward -- and the
            # AccumulateGrad nodes of every parameter under it -- stayed alive after its outputs were dropped)
            hoisted.outs = None
        return ModelOutput(last_hidden_state=hidden_states, past_key_values=next_decoder_cache,
                           hidden_states=all_hidden_states, attentions=all_self_attentions,
                           cross_attentions=all_cross_attentions)


class BertEncoderTwin(BertEncoder):
    """Two text streams: `layer` cross-attends to cat(image tokens, 3D-stream states), `layer_twin`
    to cat(object tokens, 2D-stream states); both read the OTHER stream's state of the previous
    layer (med.py:549-614)."""

    def __init__(self, config):
        super().__init__(config)
        self.num_hidden_layers_twin = getattr(config, "num_hidden_layers_twin", config.num_hidden_layers)
        self.layer_twin = nn.ModuleList([BertLayer(config, i) for i in range(self.num_hidden_layers_twin)])

    def init_twin(self):
        for i in range(self.num_hidden_layers_twin):
            self.layer_twin[i].load_state_dict(self.layer[i].state_dict())

    def _pairable(self, i, hs):
        """level i can run both streams as one stacked batch: twin layer present, kernel formats, no attention maps asked"""
        if not _TWIN_BATCH[0] or i >= self.num_hidden_layers_twin:
            return False
        a, b = self.layer[i], self.layer_twin[i]
        if a.crossattention.self.save_attention or b.crossattention.self.save_attention:
            return False
        if a.attention.self.attention_head_size != 64:
            return False
        lins = []
        for l in (a, b):
            for att in (l.attention, l.crossattention):
                lins += [att.self.query, att.self.key, att.self.value, att.output.dense]
            lins += [l.intermediate.dense, l.output.dense]
        return ops.twin_kernel_ok(hs, lins)

    def _twin_level(self, i, hs, mask2, enc2d, enc3d, mask2d, mask3d, layernorm_idx, want=False, sinks=None):
        """one level of BOTH streams on the stacked states hs (2B, L, D) (rows [0,B) = 2D stream through layer[i], rows
        [B,2B) = 3D stream through layer_twin[i]); the arithmetic per stream is BertLayer.forward's (self-attention ->
        cross-attention over cat(fixed tokens, other stream's previous states) -> FFN, post-LN; reference
        med.py:549-614), with one launch per projection / LayerNorm for the pair.  sinks: (GradSink, GradSink) of enc2d /
        enc3d -- the kernel path: the K/V projections read their two row sources in place (ops.twin_kv), nothing is
        concatenated."""
        a, b = self.layer[i], self.layer_twin[i]
        sa, sb = a.attention.self, b.attention.self
        ca, cb = a.crossattention.self, b.crossattention.self
        B2, L, D = hs.shape
        B, H, hd = B2 // 2, sa.num_attention_heads, sa.attention_head_size
        scale = 1.0 / math.sqrt(hd)
        p_att = sa.dropout.p if self.training else 0.0
        # keys / values of the cross-attentions come from the PREVIOUS states of the other stream
        if sinks is not None:
            t0 = ops.GradTap()   # (the gradient of hs through its other readers joins the K/V node's dX launch)
            kv2d, kv3d = ops.twin_kv(enc2d, enc3d, hs, (ca.key, ca.value), (cb.key, cb.value), sinks[0], sinks[1], tap=t0)
            hs = ops.tap(hs, t0)
        else:
            mix2d, mix3d = ops.twin_mix(enc2d, enc3d, hs)
        # (t1-t3: the residual-branch gradient of each sub-block's input rides on the dX GEMM of its first linear)
        t1, t2, t3 = ops.GradTap(), ops.GradTap(), ops.GradTap()
        qkv = ops.twin_multi_linear(hs, (sa.query, sa.key, sa.value), (sb.query, sb.key, sb.value), tap=t1)
        ctx = ops.attention_packed(qkv.view(B2, L, 3, H, hd), scale, p_att, mask2, return_probs=want)
        if want:
            ctx, p_self = ctx
        h = ops.twin_linear(ctx.reshape(B2, L, D), a.attention.output.dense, b.attention.output.dense)
        att = ops.twin_dropout_add_layer_norm(h, ops.tap(hs, t1), a.attention.output.LayerNorm,
                                              b.attention.output.LayerNorm, a.attention.output.dropout.p, self.training)
        q = ops.twin_linear(att, ca.query, cb.query, tap=t2).view(B2, L, H, hd)
        p_c = ca.dropout.p if self.training else 0.0
        if sinks is None:
            kv2d, kv3d = ops.twin_multi_linear_var(mix2d, mix3d, (ca.key, ca.value), (cb.key, cb.value))
        c = ops.twin_cross_attention(q, kv2d.view(B, kv2d.shape[1], 2, H, hd), kv3d.view(B, kv3d.shape[1], 2, H, hd),
                                     scale, p_c, mask2d, mask3d, return_probs=want)
        if want:
            c, p_c2d, p_c3d = c
        h = ops.twin_linear(c.reshape(B2, L, D), a.crossattention.output.dense, b.crossattention.output.dense)
        att = ops.twin_dropout_add_layer_norm(h, ops.tap(att, t2), a.crossattention.output.LayerNorm,
                                              b.crossattention.output.LayerNorm, a.crossattention.output.dropout.p,
                                              self.training)
        h = ops.twin_mlp(att, a.intermediate.dense, a.output.dense, b.intermediate.dense, b.output.dense, tap=t3)
        lna = a.output.LayerNorm if layernorm_idx == 0 else a.output.LayerNorms[layernorm_idx - 1]
        lnb = b.output.LayerNorm if layernorm_idx == 0 else b.output.LayerNorms[layernorm_idx - 1]
        out = ops.twin_dropout_add_layer_norm(h, ops.tap(att, t3), lna, lnb, a.output.dropout.p, self.training)
        if want:
            return out, (p_self[:B], p_self[B:]), (p_c2d, p_c3d)
        return out

    def forward(self, hidden_states, attention_mask=None, head_mask=None, encoder_hidden_states=None,
                encoder_attention_mask=None, encoder_hidden_states_twin=None, encoder_attention_mask_twin=None,
                past_key_values=None, use_cache=None, output_attentions=False, output_hidden_states=False,
                return_dict=True, mode="multimodal", layernorm_idx=0, forward_layers=None):
        all_hidden_states = () if output_hidden_states else None
        all_self_attentions = () if output_attentions else None
        all_cross_attentions = () if output_attentions and self.config.add_cross_attention else None
        layers = [i for i in range(self.config.num_hidden_layers) if forward_layers is None or i in forward_layers]
        hidden_states_twin = hidden_states.clone()
        ops.prime_masks(attention_mask, encoder_attention_mask, encoder_attention_mask_twin)
        enc2d = ops._c(encoder_hidden_states)
        enc3d = ops._c(encoder_hidden_states_twin)
        # kernel path: the cross-attention K/V projections read (fixed tokens, other stream's states) in place and the
        # fixed tokens' gradient accumulates in ONE buffer per stream over the levels (ops.twin_kv / ops.GradSink)
        sinks = (ops.GradSink(), ops.GradSink())
        stacked = mask2 = None   # the two streams as one (2B, L, D) tensor while consecutive levels run paired
        for i in layers:
            if output_hidden_states:
                if stacked is not None:
                    hidden_states, hidden_states_twin = ops.twin_split(stacked)
                    stacked = None
                all_hidden_states = all_hidden_states + (hidden_states,)
            want = _wants(output_attentions, i, layers[-1])
            twin = self.layer_twin[i] if i < self.num_hidden_layers_twin else None
            key_only = lambda m: m is None or (m.dim() == 4 and m.shape[1] == 1 and m.shape[2] == 1)
            if (mode == "multimodal" and attention_mask is not None
                    and key_only(attention_mask) and key_only(encoder_attention_mask)
                    and key_only(encoder_attention_mask_twin)
                    and self._pairable(i, ops._c(hidden_states) if stacked is None else stacked)):
                if stacked is None:
                    stacked = torch.cat((ops._c(hidden_states), ops._c(hidden_states_twin)), dim=0)
                if mask2 is None:
                    mask2 = getattr(attention_mask, "_bq_stacked", None)   # (BLIP_VQA3D.prepare_text: built ahead of time)
                    if mask2 is None:
                        mask2 = torch.cat((attention_mask, attention_mask), dim=0)
                    ops.prime_masks(mask2)
                stacked = self._twin_level(i, stacked, mask2, enc2d, enc3d, encoder_attention_mask,
                                           encoder_attention_mask_twin, layernorm_idx, want,
                                           sinks=sinks if ops.twin_kv_ok(enc2d, enc3d, stacked) else None)
                if want:
                    stacked, self_att, cross_att = stacked
                    all_self_attentions = all_self_attentions + (self_att,)
                    if all_cross_attentions is not None:
                        all_cross_attentions = all_cross_attentions + (cross_att,)
                continue
            if stacked is not None:
                hidden_states, hidden_states_twin = ops.twin_split(stacked)
                stacked = None
            mix2d = torch.cat((enc2d, ops._c(hidden_states_twin)), dim=1)
            mix3d = torch.cat((enc3d, ops._c(hidden_states)), dim=1)
            if twin is not None and ops.overlap_enabled(hidden_states):
                # the two streams of a layer only depend on each other's PREVIOUS state: run them side by side
                with ops.fork("twin", hidden_states) as f:
                    f.uses(hidden_states_twin, attention_mask, encoder_attention_mask_twin, mix3d)
                    out3d = twin(hidden_states_twin, attention_mask, None, mix3d, encoder_attention_mask_twin, None,
                                 want, mode=mode, layernorm_idx=layernorm_idx)
                out2d = self.layer[i](hidden_states, attention_mask, None, mix2d, encoder_attention_mask, None, want,
                                      mode=mode, layernorm_idx=layernorm_idx)
                f.join(*[t for t in out3d if isinstance(t, torch.Tensor)])
                hidden_states_twin = out3d[0]
            else:
                out2d = self.layer[i](hidden_states, attention_mask, None, mix2d, encoder_attention_mask, None, want,
                                      mode=mode, layernorm_idx=layernorm_idx)
                if twin is not None:
                    out3d = twin(hidden_states_twin, attention_mask, None, mix3d, encoder_attention_mask_twin, None,
                                 want, mode=mode, layernorm_idx=layernorm_idx)
                    hidden_states_twin = out3d[0]
            hidden_states = out2d[0]
            if want:
                self_att, cross_att = (out2d[1],), (out2d[-2],)
                if twin is not None:
                    self_att, cross_att = self_att + (out3d[1],), cross_att + (out3d[-2],)
                all_self_attentions = all_self_attentions + (self_att,)
                if all_cross_attentions is not None:
                    all_cross_attentions = all_cross_attentions + (cross_att,)
        if stacked is not None:
            hidden_states, hidden_states_twin = ops.twin_split(stacked)
        if output_hidden_states:
            all_hidden_states = all_hidden_states + (hidden_states, hidden_states_twin)
        return ModelOutput(last_hidden_state=(hidden_states, hidden_states_twin),
                           past_key_values=() if use_cache else None, hidden_states=all_hidden_states,
                           attentions=all_self_attentions, cross_attentions=all_cross_attentions)


class BertPooler(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.activation = nn.Tanh()

    def forward(self, hidden_states):
        return self.activation(self.dense(hidden_states[:, 0].float()))


class BertPredictionHeadTransform(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.transform_act_fn = ops.gelu
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)

    def forward(self, hidden_states):
        h = ops.linear(hidden_states, self.dense.weight, self.dense.bias, act="gelu")
        return ops.layer_norm(h, self.LayerNorm)


class BertLMPredictionHead(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.transform = BertPredictionHeadTransform(config)
        self.decoder = nn.Linear(config.hidden_size, config.vocab_size, bias=False)
        self.bias = nn.Parameter(torch.zeros(config.vocab_size))
        self.decoder.bias = self.bias  # same tensor under two names, as in the reference (:694)

    def forward(self, hidden_states):
        return ops.linear(self.transform(hidden_states), self.decoder.weight, self.bias)


class BertOnlyMLMHead(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.predictions = BertLMPredictionHead(config)

    def forward(self, sequence_output):
        return self.predictions(sequence_output)


class BertPreTrainedModel(nn.Module):
    """The slice of HF's PreTrainedModel this path uses: config, init_weights, dtype, mask helpers."""

    def __init__(self, config):
        super().__init__()
        self.config = config

    def _init_weights(self, module):
        if isinstance(module, (nn.Linear, nn.Embedding)):
            module.weight.data.normal_(mean=0.0, std=self.config.initializer_range)
        elif isinstance(module, nn.LayerNorm):
            module.bias.data.zero_()
            module.weight.data.fill_(1.0)
        if isinstance(module, nn.Linear) and module.bias is not None:
            module.bias.data.zero_()

    def init_weights(self):
        self.apply(self._init_weights)
        self.tie_weights()

    def tie_weights(self):
        pass

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device

    def get_extended_attention_mask(self, attention_mask, input_shape, device, is_decoder):
        """(1 - m) * -10000, causal AND padding for a decoder (med.py:771-830)."""
        if attention_mask.dim() == 3:
            ext = attention_mask[:, None, :, :]
        elif attention_mask.dim() == 2:
            if is_decoder:
                B, L = input_shape
                ids = torch.arange(L, device=device)
                causal = (ids[None, None, :].repeat(B, L, 1) <= ids[None, :, None]).to(attention_mask.dtype)
                if causal.shape[1] < attention_mask.shape[1]:
                    prefix = attention_mask.shape[1] - causal.shape[1]
                    causal = torch.cat([torch.ones((B, L, prefix), device=device, dtype=causal.dtype), causal], -1)
                ext = causal[:, None, :, :] * attention_mask[:, None, None, :]
            else:
                ext = attention_mask[:, None, None, :]
        else:
            raise ValueError("Wrong shape for input_ids (shape {}) or attention_mask (shape {})".format(
                input_shape, attention_mask.shape))
        out = (1.0 - ext.to(torch.float32)) * -10000.0
        if is_decoder and attention_mask.dim() == 2 and attention_mask.shape[1] == input_shape[1]:
            # the same mask in factored form (key padding x lower triangle) for the fused attention kernels
            out._bq_causal_key_mask = (1.0 - attention_mask[:, None, None, :].to(torch.float32)) * -10000.0
        return out

    def invert_attention_mask(self, encoder_attention_mask):
        """HF invert_attention_mask (v4.15 era): (1 - m) * -1e9 for fp32/bf16 -- not in the reference tree.
        (transformers >= 4.2x use finfo.min; equivalent whenever one key is unmasked, always true here.)"""
        if encoder_attention_mask.dim() == 3:
            ext = encoder_attention_mask[:, None, :, :]
        else:
            ext = encoder_attention_mask[:, None, None, :]
        return (1.0 - ext.to(torch.float32)) * -1e9


class BertModel(BertPreTrainedModel):
    """Encoder, or (is_decoder=True) causal decoder with cross-attention to `encoder_hidden_states`."""

    def __init__(self, config, add_pooling_layer=True):
        super().__init__(config)
        self.embeddings = BertEmbeddings(config)
        self.encoder = BertEncoder(config)
        self.pooler = BertPooler(config) if add_pooling_layer else None
        self.init_weights()

    def get_input_embeddings(self):
        return self.embeddings.word_embeddings

    def set_input_embeddings(self, value):
        self.embeddings.word_embeddings = value

    @staticmethod
    def _prepared(mask_prep, key, shape):
        """a mask computed ahead of time (extension: BLIP_VQA3D.prepare_text -> mask_prep) when it has the expected shape"""
        m = None if mask_prep is None else mask_prep.get(key)
        return m if (m is not None and tuple(m.shape) == tuple(shape)) else None

    def _prep(self, input_ids, inputs_embeds, encoder_embeds, attention_mask, past_key_values, is_decoder, mask_prep=None):
        if input_ids is not None and inputs_embeds is not None:
            raise ValueError("You cannot specify both input_ids and inputs_embeds at the same time")
        ref = input_ids if input_ids is not None else (inputs_embeds if inputs_embeds is not None else encoder_embeds)
        if ref is None:
            raise ValueError("You have to specify either input_ids or inputs_embeds or encoder_embeds")
        B, L = ref.shape[0], ref.shape[1]
        past_len = past_key_values[0][0].shape[2] if past_key_values is not None else 0
        if attention_mask is None:
            attention_mask = torch.ones((B, L + past_len), device=ref.device)
        ext = None
        if mask_prep is not None and past_len == 0:
            ext = self._prepared(mask_prep, "ext", (B, 1, L, L) if is_decoder else (B, 1, 1, L))
        if ext is None:
            ext = self.get_extended_attention_mask(attention_mask, (B, L), ref.device, is_decoder)
        return attention_mask, ext, past_len

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, head_mask=None, inputs_embeds=None,
                encoder_embeds=None, encoder_hidden_states=None, encoder_attention_mask=None, past_key_values=None,
                use_cache=None, output_attentions=None, output_hidden_states=None, return_dict=None,
                is_decoder=False, mode="multimodal", layernorm_idx=0, forward_layers=None, mask_prep=None):
        """mask_prep (extension): {"ext": this call's extended self-attention mask, "enc_ext": the inverted encoder mask}
        computed ahead of time from the token masks (BLIP_VQA3D.prepare_text); each is used when its shape fits (and, for
        enc_ext, when no encoder_attention_mask is given), otherwise the masks are built here as before"""
        output_attentions = output_attentions if output_attentions is not None else self.config.output_attentions
        output_hidden_states = output_hidden_states if output_hidden_states is not None \
            else self.config.output_hidden_states
        use_cache = (use_cache if use_cache is not None else self.config.use_cache) if is_decoder else False
        attention_mask, ext_mask, past_len = self._prep(input_ids, inputs_embeds, encoder_embeds, attention_mask,
                                                        past_key_values, is_decoder, mask_prep)
        enc_ext = None
        if encoder_hidden_states is not None:
            if encoder_attention_mask is None:
                enc_ext = self._prepared(mask_prep, "enc_ext", (encoder_hidden_states.shape[0], 1, 1, encoder_hidden_states.shape[1]))
                if enc_ext is None:
                    encoder_attention_mask = torch.ones(encoder_hidden_states.shape[:2], device=ext_mask.device)
            if enc_ext is None:
                enc_ext = self.invert_attention_mask(encoder_attention_mask)
        if encoder_embeds is None:
            embedding_output = self.embeddings(input_ids=input_ids, position_ids=position_ids,
                                               inputs_embeds=inputs_embeds, past_key_values_length=past_len)
        else:
            embedding_output = encoder_embeds
        enc = self.encoder(embedding_output, attention_mask=ext_mask, encoder_hidden_states=encoder_hidden_states,
                           encoder_attention_mask=enc_ext, past_key_values=past_key_values, use_cache=use_cache,
                           output_attentions=output_attentions, output_hidden_states=output_hidden_states,
                           mode=mode, layernorm_idx=layernorm_idx, forward_layers=forward_layers)
        seq = enc.last_hidden_state
        pooled = self.pooler(seq) if self.pooler is not None else None
        return ModelOutput(last_hidden_state=seq, pooler_output=pooled, past_key_values=enc.past_key_values,
                           hidden_states=enc.hidden_states, attentions=enc.attentions,
                           cross_attentions=enc.cross_attentions)


class BertModelTwin(BertModel):
    """BertModel whose encoder is the twin encoder; last_hidden_state is the pair (h2d, h3d)."""

    def __init__(self, config, add_pooling_layer=True):
        super().__init__(config)  # NB: like the reference, the base ctor always builds a pooler first (:976-983)
        self.config = config
        self.encoder = BertEncoderTwin(config)
        self.pooler = BertPooler(config) if add_pooling_layer else None
        self.pooler_twin = BertPooler(config) if add_pooling_layer else None

    def init_twin(self):
        self.encoder.init_twin()
        if self.pooler is not None:
            self.pooler_twin.load_state_dict(self.pooler.state_dict())

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, head_mask=None, inputs_embeds=None,
                encoder_embeds=None, encoder_hidden_states=None, encoder_attention_mask=None,
                encoder_hidden_states_twin=None, encoder_attention_mask_twin=None, past_key_values=None,
                use_cache=None, output_attentions=None, output_hidden_states=None, return_dict=None,
                is_decoder=False, mode="multimodal", layernorm_idx=0, forward_layers=None, mask_prep=None):
        """mask_prep (extension, see BertModel.forward): "ext" and, for an all-ones encoder mask (encoder_attention_mask
        None: the image tokens), "enc_ext" = inverted cat(ones, text mask)"""
        output_attentions = output_attentions if output_attentions is not None else self.config.output_attentions
        output_hidden_states = output_hidden_states if output_hidden_states is not None \
            else self.config.output_hidden_states
        attention_mask, ext_mask, past_len = self._prep(input_ids, inputs_embeds, encoder_embeds, attention_mask,
                                                        past_key_values, is_decoder, mask_prep)
        # each stream also sees the other stream's text states: masks are cat(encoder mask, text mask)
        enc_ext = None
        if encoder_attention_mask is None:
            L1 = encoder_hidden_states.shape[1] + attention_mask.shape[1]
            enc_ext = self._prepared(mask_prep, "enc_ext", (attention_mask.shape[0], 1, 1, L1))
            if enc_ext is None:
                encoder_attention_mask = torch.ones(encoder_hidden_states.shape[:2], dtype=torch.long, device=ext_mask.device)
        if enc_ext is None:
            am = attention_mask.to(encoder_attention_mask.dtype)
            enc_ext = self.invert_attention_mask(torch.cat((encoder_attention_mask, am), dim=1))
        am = attention_mask.to(encoder_attention_mask_twin.dtype)
        enc_ext_twin = self.invert_attention_mask(torch.cat((encoder_attention_mask_twin, am), dim=1))
        if encoder_embeds is None:
            embedding_output = self.embeddings(input_ids=input_ids, position_ids=position_ids,
                                               inputs_embeds=inputs_embeds, past_key_values_length=past_len)
        else:
            embedding_output = encoder_embeds
        enc = self.encoder(embedding_output, attention_mask=ext_mask, encoder_hidden_states=encoder_hidden_states,
                           encoder_attention_mask=enc_ext, encoder_hidden_states_twin=encoder_hidden_states_twin,
                           encoder_attention_mask_twin=enc_ext_twin, output_attentions=output_attentions,
                           output_hidden_states=output_hidden_states, mode=mode, layernorm_idx=layernorm_idx,
                           forward_layers=forward_layers)
        h2d, h3d = enc.last_hidden_state
        pooled = self.pooler(h2d) if self.pooler is not None else None
        pooled_twin = self.pooler(h3d) if self.pooler is not None else None  # sic: reference uses `pooler` twice
        return ModelOutput(last_hidden_state=(h2d, h3d), pooler_output=(pooled, pooled_twin),
                           past_key_values=enc.past_key_values, hidden_states=enc.hidden_states,
                           attentions=enc.attentions, cross_attentions=enc.cross_attentions)


class BertLMHeadModel(BertPreTrainedModel):
    """Causal answer decoder with the LM head tied to the word embeddings; label-smoothed CE."""

    def __init__(self, config):
        super().__init__(config)
        self.bert = BertModel(config, add_pooling_layer=False)
        self.cls = BertOnlyMLMHead(config)
        self.init_weights()

    def get_output_embeddings(self):
        return self.cls.predictions.decoder

    def set_output_embeddings(self, new_embeddings):
        self.cls.predictions.decoder = new_embeddings

    def tie_weights(self):
        # HF PreTrainedModel.tie_weights: output embedding weight IS the input embedding weight
        self.cls.predictions.decoder.weight = self.bert.embeddings.word_embeddings.weight

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, head_mask=None, inputs_embeds=None,
                encoder_hidden_states=None, encoder_attention_mask=None, labels=None, past_key_values=None,
                use_cache=None, output_attentions=None, output_hidden_states=None, return_dict=None,
                return_logits=False, is_decoder=True, reduction="mean", mode="multimodal", layernorm_idx=0,
                encoder_embeds=None, mask_prep=None):
        """encoder_embeds (extension): the output of self.bert.embeddings computed ahead of time -- the text prologue of
        pipeline.PhasedTrainStep (BLIP_VQA3D.prepare_text); BertModel.forward takes it as the reference's does"""
        if labels is not None:
            use_cache = False
        outputs = self.bert(input_ids, attention_mask=attention_mask, position_ids=position_ids,
                            inputs_embeds=inputs_embeds, encoder_embeds=encoder_embeds,
                            encoder_hidden_states=encoder_hidden_states,
                            encoder_attention_mask=encoder_attention_mask, past_key_values=past_key_values,
                            use_cache=use_cache, output_attentions=output_attentions,
                            output_hidden_states=output_hidden_states, is_decoder=is_decoder, mode=mode,
                            layernorm_idx=layernorm_idx, mask_prep=mask_prep)
        head = self.cls.predictions
        hidden = head.transform(outputs.last_hidden_state)
        lm_loss = None
        if labels is not None and not return_logits:
            logits, per_seq = ops.lm_loss(hidden, head.decoder.weight, head.bias, labels, label_smoothing=0.1)
            if reduction == "none":
                lm_loss = per_seq
            else:
                n = (labels[:, 1:] != -100).sum().clamp(min=1)
                lm_loss = per_seq.sum() / n if reduction == "mean" else per_seq.sum()
        else:
            logits = ops.linear(hidden, head.decoder.weight, head.bias)
        if return_logits:
            return logits[:, :-1, :].contiguous()
        return ModelOutput(loss=lm_loss, logits=logits, past_key_values=outputs.past_key_values,
                           hidden_states=outputs.hidden_states, attentions=outputs.attentions,
                           cross_attentions=outputs.cross_attentions)

    # ---- generation (reference med.py:1447-1470 + HF GenerationMixin.generate, see generation.py) ----------------------
    def prepare_inputs_for_generation(self, input_ids, past=None, attention_mask=None, **model_kwargs):
        if attention_mask is None:
            attention_mask = input_ids.new_ones(input_ids.shape)
        if past is not None:
            input_ids = input_ids[:, -1:]
        return {"input_ids": input_ids, "attention_mask": attention_mask, "past_key_values": past,
                "encoder_hidden_states": model_kwargs.get("encoder_hidden_states", None),
                "encoder_attention_mask": model_kwargs.get("encoder_attention_mask", None), "is_decoder": True}

    def _reorder_cache(self, past, beam_idx):
        return tuple(tuple(t.index_select(0, beam_idx) for t in layer_past) for layer_past in past)

    @torch.no_grad()
    def generate(self, input_ids, max_length=20, min_length=0, num_beams=1, eos_token_id=None, pad_token_id=None,
                 length_penalty=1.0, early_stopping=False, return_scores=False, **model_kwargs):
        """beam search as `transformers` v4.15 `generate` runs it for a decoder-only model (generation.py; parity
        unpinned): input_ids (B, L0) is repeated num_beams times per sample; encoder_hidden_states / encoder_attention_mask
        must already have B * num_beams rows (the caller chooses what every beam slot attends to)."""
        from .generation import beam_search
        if eos_token_id is None or pad_token_id is None:
            raise ValueError("generate needs eos_token_id and pad_token_id")
        ids = input_ids.repeat_interleave(num_beams, dim=0)
        enc = model_kwargs.get("encoder_hidden_states")
        if enc is not None and enc.shape[0] != ids.shape[0]:
            raise ValueError("encoder_hidden_states must have batch * num_beams = %d rows, got %d" % (ids.shape[0], enc.shape[0]))

        def step(cur, past):
            inp = self.prepare_inputs_for_generation(cur, past=past, **model_kwargs)
            out = self(**inp, use_cache=True, return_dict=True)
            return out.logits[:, -1, :], out.past_key_values

        seq, scores = beam_search(step, self._reorder_cache, ids, num_beams, max_length, eos_token_id, pad_token_id,
                                  min_length=min_length, length_penalty=length_penalty, early_stopping=early_stopping)
        return (seq, scores) if return_scores else seq

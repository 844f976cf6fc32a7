"""ViT-B/16 image encoder -- mirror of the reference's models/vit.py:23-196,283-307 and of the
two timm pieces it imports (PatchEmbed = Conv2d(3,D,16,16) -> flatten -> transpose; DropPath =
per-sample Bernoulli(keep)/keep), which are NOT vendored in the reference tree (timm 0.4.12).

State-dict names: cls_token, pos_embed, patch_embed.proj.{weight,bias},
blocks.{i}.{norm1,norm2}.{weight,bias}, blocks.{i}.attn.{qkv,proj}.{weight,bias},
blocks.{i}.mlp.{fc1,fc2}.{weight,bias}, norm.{weight,bias}.

Compute goes through bridgeqa_amd.fusion_ops (bf16-in / fp32-accumulate GEMMs with fused
epilogues and a non-materialising attention) -- the attention probability tensor
(B,12,P,P) of vit.py:75-83 is never written to HBM.
"""
from functools import partial

import torch
import torch.nn as nn

from . import fusion_ops as ops


class PatchEmbed(nn.Module):
    """timm.models.vision_transformer.PatchEmbed (used at vit.py:144-145,182)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        img_size = (img_size, img_size) if isinstance(img_size, int) else tuple(img_size)
        patch_size = (patch_size, patch_size) if isinstance(patch_size, int) else tuple(patch_size)
        self.img_size, self.patch_size = img_size, patch_size
        self.grid_size = (img_size[0] // patch_size[0], img_size[1] // patch_size[1])
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

    def forward(self, x):
        B, C, H, W = x.shape
        assert H == self.img_size[0] and W == self.img_size[1], \
            "Input image size (%d*%d) doesn't match model (%d*%d)." % (H, W, self.img_size[0], self.img_size[1])
        # conv with stride == kernel  ==  GEMM over unfolded 16x16x3 patches
        ph, pw = self.patch_size
        gh, gw = self.grid_size
        patches = x.reshape(B, C, gh, ph, gw, pw).permute(0, 2, 4, 1, 3, 5).reshape(B, gh * gw, C * ph * pw)
        return ops.linear(patches, self.proj.weight, self.proj.bias)  # (N, C, kh, kw) read as the (N, C*kh*kw) GEMM operand


class DropPath(nn.Module):
    """timm.models.layers.DropPath: stochastic depth per sample, identity in eval / p=0."""

    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.dim() - 1)).bernoulli_(keep)
        return x.div(keep) * mask


class Mlp(nn.Module):
    """fc1 -> exact GELU -> fc2   (vit.py:23-41; dropout p=0 on this path)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)

    def forward(self, x):
        if self.drop.p == 0.0 or not self.training:
            return ops.mlp(x, self.fc1, self.fc2)  # one autograd node: GELU / dGELU / fc1's bias gradient in epilogues
        x = ops.linear(x, self.fc1.weight, self.fc1.bias, act="gelu")
        x = self.drop(x)
        x = ops.linear(x, self.fc2.weight, self.fc2.bias)
        return self.drop(x)


class Attention(nn.Module):
    """Fused-QKV multi-head self-attention (vit.py:44-86)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = qk_scale or head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self.attn_gradients = None
        self.attention_map = None

    def save_attn_gradients(self, attn_gradients):
        self.attn_gradients = attn_gradients

    def get_attn_gradients(self):
        return self.attn_gradients

    def save_attention_map(self, attention_map):
        self.attention_map = attention_map

    def get_attention_map(self):
        return self.attention_map

    def forward(self, x, register_hook=False):
        B, N, C = x.shape
        H = self.num_heads
        qkv = ops.linear(x, self.qkv.weight, self.qkv.bias).reshape(B, N, 3, H, C // H)
        drop = self.attn_drop.p if self.training else 0.0
        if register_hook:  # Grad-CAM style hooks need the probabilities: reference composition
            ctx, probs = ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], None, self.scale, return_probs=True,
                                       dropout_p=drop)
            self.save_attention_map(probs)
            if probs.requires_grad:
                probs.register_hook(self.save_attn_gradients)
        else:
            ctx = ops.attention_packed(qkv, self.scale, dropout_p=drop)
        x = ops.linear(ctx.reshape(B, N, C), self.proj.weight, self.proj.bias)
        return self.proj_drop(x)


class Block(nn.Module):
    """Pre-LN transformer block (vit.py:89-110)."""

    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, qk_scale=None, drop=0.0, attn_drop=0.0,
                 drop_path=0.0, act_layer=nn.GELU, norm_layer=nn.LayerNorm, use_grad_checkpointing=False):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop,
                              proj_drop=drop)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        self.use_grad_checkpointing = use_grad_checkpointing  # fairscale checkpoint_wrapper in the reference

    def _attn(self, x, register_hook):
        return self.attn(ops.layer_norm(x, self.norm1), register_hook=register_hook)

    def _mlp(self, x):
        return self.mlp(ops.layer_norm(x, self.norm2))

    def forward(self, x, register_hook=False):
        if self.use_grad_checkpointing and self.training and not register_hook:
            from torch.utils.checkpoint import checkpoint
            x = x + self.drop_path(checkpoint(self._attn, x, False, use_reentrant=False))
            x = x + self.drop_path(checkpoint(self._mlp, x, use_reentrant=False))
            return x
        x = x + self.drop_path(self._attn(x, register_hook))
        x = x + self.drop_path(self._mlp(x))
        return x

    def fusable(self, register_hook=False):
        """no checkpointing, no hooks: the block can run as forward_fused with each residual add (and its
        stochastic depth) folded into the following LayerNorm"""
        return not register_hook and not (self.use_grad_checkpointing and self.training)

    def forward_fused(self, x, n1, next_norm):
        """x: residual stream, n1 = norm1(x) (already computed).  Returns (x_out, next_norm(x_out))."""
        dp = self.drop_path.drop_prob if (self.training and isinstance(self.drop_path, DropPath)) else 0.0
        x, n2 = ops.add_layer_norm(self.attn(n1), x, self.norm2, drop_path=dp)
        return ops.add_layer_norm(self.mlp(n2), x, next_norm, drop_path=dp)


class VisionTransformer(nn.Module):
    """vit.py:113-196.  forward(x (B,3,H,W)) -> (B, 1 + H*W/256, embed_dim)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4.0, qkv_bias=True, qk_scale=None, representation_size=None,
                 drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.0, norm_layer=None,
                 use_grad_checkpointing=False, ckpt_layer=0):
        super().__init__()
        self.num_features = self.embed_dim = embed_dim
        norm_layer = norm_layer or partial(nn.LayerNorm, eps=1e-6)
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans,
                                      embed_dim=embed_dim)
        num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, embed_dim))
        self.pos_drop = nn.Dropout(p=drop_rate)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]  # stochastic depth decay rule
        self.blocks = nn.ModuleList([
            Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                  drop=drop_rate, attn_drop=attn_drop_rate, drop_path=dpr[i], norm_layer=norm_layer,
                  use_grad_checkpointing=(use_grad_checkpointing and i >= depth - ckpt_layer))
            for i in range(depth)])
        self.norm = norm_layer(embed_dim)
        # autograd cuts (pipeline.PhasedTrainStep(image_bwd_splits=...)): scoped to ONE forward by autograd_cuts() --
        # never persistent state, a plain forward + loss.backward() must reach patch_embed
        self.grad_cuts, self.cut_pairs = (), []
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        nn.init.trunc_normal_(self.cls_token, std=0.02)
        self.apply(self._init_weights)

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {"pos_embed", "cls_token"}

    def autograd_cuts(self, cuts):
        """context manager: forwards run inside it detach the residual stream in front of blocks `cuts` and leave the
        (outputs, leaves) pairs in self.cut_pairs; outside it grad_cuts is () again"""
        return _AutogradCuts(self, cuts)

    def forward(self, x, register_blk=-1, return_fm=-1):
        B = x.shape[0]
        self.cut_pairs = []
        x = self.patch_embed(x)
        x = torch.cat((self.cls_token.expand(B, -1, -1).to(x.dtype), x), dim=1)
        x = x + self.pos_embed[:, :x.size(1), :].to(x.dtype)
        x = self.pos_drop(x)
        last = len(self.blocks) - 1
        if return_fm == -1 and all(blk.fusable(register_blk == i) for i, blk in enumerate(self.blocks)):
            # same arithmetic, 2 launches per block instead of 2 adds + 2 norms (+ casts)
            n = ops.layer_norm(x, self.blocks[0].norm1)
            cuts = self.grad_cuts
            for i, blk in enumerate(self.blocks):
                if i in cuts and torch.is_grad_enabled() and x.requires_grad:
                    # autograd cut in front of block i (pipeline.PhasedTrainStep(image_bwd_splits=...)): the backward then
                    # runs in block ranges, last range first -- ((x, n) outputs of the earlier range, the leaves the later
                    # range consumed): torch.autograd.backward([x, n], [xl.grad, nl.grad]) continues the chain
                    xl, nl = x.detach().requires_grad_(True), n.detach().requires_grad_(True)
                    self.cut_pairs.append(((x, n), (xl, nl)))
                    x, n = xl, nl
                x, n = blk.forward_fused(x, n, self.blocks[i + 1].norm1 if i < last else self.norm)
            return n
        for i, blk in enumerate(self.blocks):
            x = blk(x, register_blk == i)
            if len(self.blocks) + return_fm == i:
                break
        return ops.layer_norm(x, self.norm)


class _AutogradCuts(object):
    def __init__(self, vit, cuts):
        self.vit, self.cuts = vit, tuple(cuts)

    def __enter__(self):
        self.prev, self.vit.grad_cuts = self.vit.grad_cuts, self.cuts
        return self.vit

    def __exit__(self, *exc):
        self.vit.grad_cuts = self.prev
        return False


def interpolate_pos_embed(pos_embed_checkpoint, visual_encoder):
    """Bicubic resize of a checkpoint's position table to this encoder's grid (vit.py:283-307)."""
    embedding_size = pos_embed_checkpoint.shape[-1]
    num_patches = visual_encoder.patch_embed.num_patches
    num_extra_tokens = visual_encoder.pos_embed.shape[-2] - num_patches
    orig_size = int((pos_embed_checkpoint.shape[-2] - num_extra_tokens) ** 0.5)
    new_size = int(num_patches ** 0.5)
    if orig_size == new_size:
        return pos_embed_checkpoint
    extra_tokens = pos_embed_checkpoint[:, :num_extra_tokens]
    pos_tokens = pos_embed_checkpoint[:, num_extra_tokens:]
    pos_tokens = pos_tokens.reshape(-1, orig_size, orig_size, embedding_size).permute(0, 3, 1, 2)
    pos_tokens = torch.nn.functional.interpolate(pos_tokens, size=(new_size, new_size), mode="bicubic",
                                                 align_corners=False)
    pos_tokens = pos_tokens.permute(0, 2, 3, 1).flatten(1, 2)
    return torch.cat((extra_tokens, pos_tokens), dim=1)


def create_vit(vit, image_size, use_grad_checkpointing=False, ckpt_layer=0, drop_path_rate=0):
    """models/blip.py:334-365."""
    assert vit in ["base", "large"], "vit parameter must be base or large"
    if vit == "base":
        width = 768
        enc = VisionTransformer(img_size=image_size, patch_size=16, embed_dim=width, depth=12, num_heads=12,
                                use_grad_checkpointing=use_grad_checkpointing, ckpt_layer=ckpt_layer,
                                drop_path_rate=0 or drop_path_rate)
    else:
        width = 1024
        enc = VisionTransformer(img_size=image_size, patch_size=16, embed_dim=width, depth=24, num_heads=16,
                                use_grad_checkpointing=use_grad_checkpointing, ckpt_layer=ckpt_layer,
                                drop_path_rate=0.1 or drop_path_rate)
    return enc, width

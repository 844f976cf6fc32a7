"""vit.VisionTransformer.grad_cuts: the autograd cuts pipeline.PhasedTrainStep(image_bwd_splits=...) uses to run the image
encoder's backward in block ranges (each range's gradients are exchanged while the next range runs -- what DDP's buckets
firing during backward do, scripts/train.py:346-347).  The cut forward is the same function, and the backward continued
range by range produces the same gradients as one backward."""
import torch


def _vit():
    from bridgeqa_amd.vit import VisionTransformer
    torch.manual_seed(0)
    m = VisionTransformer(img_size=32, patch_size=16, embed_dim=64, depth=6, num_heads=1, drop_path_rate=0.0)
    return m.train()


def test_block_range_backward_equals_one_backward():
    m = _vit()
    x = torch.randn(2, 3, 32, 32)
    w = torch.randn(2, 5, 64)
    y0 = m(x)
    (y0 * w).sum().backward()
    want = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    for p in m.parameters():
        p.grad = None
    with m.autograd_cuts((2, 4)):
        y1 = m(x)
    assert m.grad_cuts == ()                                  # scoped to the forward inside the context
    assert torch.equal(y0, y1) and len(m.cut_pairs) == 2
    y1.backward(w)                                            # blocks 4-5 (+ final norm, + block 4's norm1 owner: block 3)
    seen = [{n for n, p in m.named_parameters() if p.grad is not None}]
    for (xo, no), (xl, nl) in reversed(m.cut_pairs):          # blocks 2-3, then 0-1 + embeddings
        torch.autograd.backward([xo, no], [xl.grad, nl.grad])
        seen.append({n for n, p in m.named_parameters() if p.grad is not None})
    got = {n: p.grad for n, p in m.named_parameters() if p.grad is not None}
    assert set(got) == set(want)
    for n in want:
        assert torch.allclose(got[n], want[n], rtol=1e-5, atol=1e-6), n
    # every range produced gradients of its own, the first range (run last) the embeddings'
    assert seen[0] < seen[1] < seen[2]
    assert "blocks.5.attn.qkv.weight" in seen[0] and "blocks.0.attn.qkv.weight" not in seen[1]
    assert "patch_embed.proj.weight" in seen[2] - seen[1]
    # a block's norm1 is applied by the block BEFORE it (fused add + LayerNorm): its gradient comes from that range
    assert "blocks.4.norm1.weight" in seen[1] - seen[0]


def test_no_cut_without_grad():
    m = _vit()
    with m.autograd_cuts((2,)), torch.no_grad():
        m(torch.randn(1, 3, 32, 32))
    assert m.cut_pairs == []


def test_cuts_are_scoped_and_a_plain_backward_reaches_patch_embed():
    """the cuts were persistent module state -- after a cut forward, an ordinary forward + backward must
    differentiate down to the embeddings again, and a per-block (non fused) forward leaves no stale pairs"""
    m = _vit()
    x = torch.randn(2, 3, 32, 32)
    with m.autograd_cuts((2, 4)):
        m(x)
    assert len(m.cut_pairs) == 2
    m(x).sum().backward()
    assert m.cut_pairs == [] and m.grad_cuts == ()
    assert m.patch_embed.proj.weight.grad is not None and m.pos_embed.grad is not None
    assert m.blocks[0].attn.qkv.weight.grad.abs().sum() > 0
    with m.autograd_cuts((2,)):
        m(x)
        assert len(m.cut_pairs) == 1
        m(x, return_fm=-2)                                    # per-block path: no cuts, and no stale ones either
        assert m.cut_pairs == []

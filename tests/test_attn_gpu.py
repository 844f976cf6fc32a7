"""Fused attention kernel (csrc/attn.hip) against a plain PyTorch fp32 reference of the same op
(softmax(q k^T * scale) v on the same bf16-rounded inputs).  Tolerance: bf16 probabilities / outputs,
fp32 accumulation -> 2e-2 relative-L2 on the output, 1e-3 absolute on the LSE."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def ref_attention(q, k, v, scale):
    qf, kf, vf = (t.float().permute(0, 2, 1, 3) for t in (q, k, v))
    s = torch.matmul(qf, kf.transpose(-1, -2)) * scale
    p = torch.softmax(s, dim=-1)
    return torch.matmul(p, vf).permute(0, 2, 1, 3), torch.logsumexp(s, dim=-1) / math.log(2.0)


@pytest.mark.parametrize("B,H,L", [(1, 1, 32), (2, 3, 64), (1, 2, 77), (2, 12, 197), (1, 4, 1025), (2, 2, 901),
                                   (1, 1, 1), (1, 2, 129), (1, 2, 4097)])  # 4097 = config c5 (1024^2 view)
def test_attn_fwd_vs_torch_fp32(dev, B, H, L):
    from bridgeqa_amd import _ext
    g = torch.Generator().manual_seed(L)
    qkv = (torch.randn(B, L, 3, H, 64, generator=g) * 1.5).to(dev).to(torch.bfloat16)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]  # strided views, as the fused-QKV GEMM produces them
    out, lse = _ext.attn_fwd(q, k, v, 0.125)
    want, want_lse = ref_attention(q, k, v, 0.125)
    err = (out.float() - want).norm() / want.norm()
    assert err < 2e-2, err
    assert (lse - want_lse).abs().max() < 2e-3


@pytest.mark.parametrize("B,H,L", [(16, 12, 1025), (16, 12, 1024), (16, 12, 1100), (32, 12, 257), (64, 12, 129),
                                   (16, 12, 1026)])
def test_attn_fwd_resident_grid_vs_per_block_and_fp32(dev, B, H, L):
    """the resident-grid forward (attn_fwd_persist_kernel: item walk, cross-item prefetch, the 128 n + 1-th row on vector
    arithmetic) against the one-workgroup-per-block launch on the same inputs and against fp32 torch; shapes: the ViT's
    (one ragged row), whole blocks only, a ragged block of 76 rows, of 2 rows, and two short ones with one ragged row"""
    from bridgeqa_amd import _ext
    g = torch.Generator().manual_seed(L + B)
    qkv = (torch.randn(B, L, 3, H, 64, generator=g) * 1.5).to(dev).to(torch.bfloat16)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    prev = _ext.attn_set_persistent(0)
    try:
        base, base_lse = _ext.attn_fwd(q, k, v, 0.125)
        _ext.attn_set_persistent(7)
        out, lse = _ext.attn_fwd(q, k, v, 0.125)
    finally:
        _ext.attn_set_persistent(prev)
    assert torch.isfinite(out.float()).all() and torch.isfinite(lse).all()
    # the same products on the matrix path; the softmax sums pair differently (packed fp32 adds) and the vector-arithmetic
    # row rounds in another order: a few bf16 ulps of the output
    d = (out.float() - base.float()).abs().max().item()
    assert d < 2e-2, d
    assert (lse - base_lse).abs().max() < 1e-3
    assert (out.float() - base.float()).norm() / base.float().norm() < 2e-3
    for b in (0, B - 1):
        want, want_lse = ref_attention(q[b:b + 1], k[b:b + 1], v[b:b + 1], 0.125)
        err = (out[b:b + 1].float() - want).norm() / want.norm()
        assert err < 2e-2, err
        assert (lse[b:b + 1] - want_lse).abs().max() < 2e-3
        tail = (out[b, L - 1].float() - want[0, L - 1]).norm() / want[0, L - 1].norm()
        assert tail < 2e-2, tail


@pytest.mark.parametrize("B,H,L", [(16, 12, 1025), (16, 12, 1024), (16, 12, 1100), (32, 12, 257), (16, 12, 1026)])
def test_attn_bwd_resident_grid_vs_per_block_and_fp32(dev, B, H, L):
    """the resident-grid dQ and dK/dV passes (item walk, cross-item tile prefetch, the 128 n + 1-th row / key on vector
    arithmetic) against the block-by-block launches on the same inputs (bit-equal on the matrix path) and, for two batch
    elements, against fp32 autograd -- the ragged row / key on its own"""
    from bridgeqa_amd import _ext
    g = torch.Generator().manual_seed(3 * L + B)
    qkv = torch.randn(B, L, 3, H, 64, generator=g).to(dev).to(torch.bfloat16)
    go = torch.randn(B, L, H, 64, generator=g).to(dev).to(torch.bfloat16)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    prev = _ext.attn_set_persistent(0)
    try:
        out, lse = _ext.attn_fwd(q, k, v, 0.125)
        base = torch.zeros_like(qkv)
        _ext.attn_bwd(q, k, v, out, lse, go, 0.125, base[:, :, 0], base[:, :, 1], base[:, :, 2])
        _ext.attn_set_persistent(7)
        got = torch.zeros_like(qkv)
        _ext.attn_bwd(q, k, v, out, lse, go, 0.125, got[:, :, 0], got[:, :, 1], got[:, :, 2])
    finally:
        _ext.attn_set_persistent(prev)
    assert torch.isfinite(got.float()).all()
    n = L - 1 if L % 128 == 1 else L
    assert torch.equal(got[:, :n], base[:, :n])
    rel = lambda a, b: float((a - b).norm() / b.norm())
    if n < L:
        for i, name in enumerate(("dq", "dk", "dv")):
            assert rel(got[:, n:, i].float(), base[:, n:, i].float()) < 1e-2, name
    for b in (0, B - 1):
        ref_in = qkv[b:b + 1].float().requires_grad_(True)
        want, _ = ref_attention(ref_in[:, :, 0], ref_in[:, :, 1], ref_in[:, :, 2], 0.125)
        want.backward(go[b:b + 1].float())
        for i, name in enumerate(("dq", "dk", "dv")):
            assert rel(got[b:b + 1, :, i].float(), ref_in.grad[:, :, i]) < 3e-2, name
            assert rel(got[b:b + 1, L - 1:, i].float(), ref_in.grad[:, L - 1:, i]) < 3e-2, (name, "last row")


def test_attn_fwd_rescale_branch(dev):
    """Force the online-softmax running max to jump late (a spiked key at the end of the sequence)."""
    from bridgeqa_amd import _ext
    g = torch.Generator().manual_seed(0)
    B, H, L = 1, 2, 300
    q = torch.randn(B, L, H, 64, generator=g)
    k = torch.randn(B, L, H, 64, generator=g)
    v = torch.randn(B, L, H, 64, generator=g)
    k[:, 290] = q[:, 5] * 4.0  # one key strongly aligned with one query, in the last tile
    q, k, v = (t.to(dev).to(torch.bfloat16) for t in (q, k, v))
    out, lse = _ext.attn_fwd(q, k, v, 0.125)
    want, want_lse = ref_attention(q, k, v, 0.125)
    assert (out.float() - want).norm() / want.norm() < 2e-2
    assert (lse - want_lse).abs().max() < 2e-3


@pytest.mark.parametrize("B,H,L", [(1, 1, 32), (2, 3, 64), (1, 2, 77), (2, 4, 197), (1, 2, 1025), (1, 1, 130),
                                   (1, 1, 4097)])
def test_attn_bwd_vs_torch_autograd_fp32(dev, B, H, L):
    """dQ, dK, dV of the fused kernels vs autograd through the fp32 reference composition on the same
    bf16-rounded inputs; tolerance 3e-2 relative-L2 (bf16 P / dS operands, fp32 accumulation)."""
    from bridgeqa_amd import fusion_ops
    g = torch.Generator().manual_seed(L + 1)
    qkv = (torch.randn(B, L, 3, H, 64, generator=g)).to(dev).to(torch.bfloat16).requires_grad_(True)
    go = torch.randn(B, L, H, 64, generator=g).to(dev).to(torch.bfloat16)
    out = fusion_ops.attention_packed(qkv, 0.125)
    out.backward(go)
    got = qkv.grad.float()
    ref_in = qkv.detach().float().requires_grad_(True)
    want, _ = ref_attention(ref_in[:, :, 0], ref_in[:, :, 1], ref_in[:, :, 2], 0.125)
    want.backward(go.float())
    for i, name in enumerate(("dq", "dk", "dv")):
        err = (got[:, :, i] - ref_in.grad[:, :, i]).norm() / ref_in.grad[:, :, i].norm()
        assert err < 3e-2, (name, err)


@pytest.mark.parametrize("B,H,L", [(2, 3, 129), (1, 2, 257), (1, 2, 1025), (2, 2, 160), (1, 2, 161)])
def test_attn_ragged_last_block_rows_and_keys(dev, B, H, L):
    """L = 128 n + r (a ViT's class token: r = 1): the rows / keys of the ragged last block are 1 in ~1000 of the tensors
    the whole-tensor tests above norm over, so they are checked on their own here."""
    from bridgeqa_amd import fusion_ops, _ext
    g = torch.Generator().manual_seed(7 * L)
    qkv = (torch.randn(B, L, 3, H, 64, generator=g)).to(dev).to(torch.bfloat16).requires_grad_(True)
    go = torch.randn(B, L, H, 64, generator=g).to(dev).to(torch.bfloat16)
    out = fusion_ops.attention_packed(qkv, 0.125)
    out.backward(go)
    _, lse = _ext.attn_fwd(qkv.detach()[:, :, 0], qkv.detach()[:, :, 1], qkv.detach()[:, :, 2], 0.125)
    ref_in = qkv.detach().float().requires_grad_(True)
    want, want_lse = ref_attention(ref_in[:, :, 0], ref_in[:, :, 1], ref_in[:, :, 2], 0.125)
    want.backward(go.float())
    r = L % 128 if L % 128 else 128
    rows = slice(L - r, L)
    rel = lambda a, b: float((a - b).norm() / b.norm())
    assert rel(out.detach().float()[:, rows], want.detach()[:, rows]) < 2e-2
    assert (lse[:, :, rows] - want_lse.detach()[:, :, rows]).abs().max() < 2e-3
    got = qkv.grad.float()
    for i, name in enumerate(("dq", "dk", "dv")):
        e_tail = rel(got[:, rows, i], ref_in.grad[:, rows, i])
        e_rest = rel(got[:, :L - r, i], ref_in.grad[:, :L - r, i])
        assert e_tail < 3e-2 and e_rest < 3e-2, (name, e_tail, e_rest)


def test_vit_block_bf16_fused_vs_fp32_composition(dev):
    """A ViT block through the fused path (bf16) against the same block in fp32 reference composition."""
    from bridgeqa_amd import fusion_ops, vit
    torch.manual_seed(0)
    blk = vit.Block(dim=768, num_heads=12, qkv_bias=True).to(dev).eval()
    x = torch.randn(2, 197, 768, device=dev)
    prev = fusion_ops.set_compute_dtype(torch.float32)
    try:
        want = blk(x)
        fusion_ops.set_compute_dtype(torch.bfloat16)
        got = blk(x)
    finally:
        fusion_ops.set_compute_dtype(prev)
    assert (got.float() - want).norm() / want.norm() < 2e-2


@pytest.mark.parametrize("act", [None, "gelu"])
def test_bf16_linear_with_fp32_master_weights(dev, act):
    """fusion_ops.linear in bf16 mode: output and fp32 gradients vs the fp32 composition (tolerance = bf16 operands)."""
    from bridgeqa_amd import fusion_ops
    torch.manual_seed(0)
    lin = torch.nn.Linear(96, 160).to(dev)
    x = torch.randn(4, 33, 96, device=dev, requires_grad=True)
    prev = fusion_ops.set_compute_dtype(torch.bfloat16)
    try:
        y = fusion_ops.linear(x, lin.weight, lin.bias, act=act)
        assert y.dtype == torch.bfloat16
        y.float().square().sum().backward()
    finally:
        fusion_ops.set_compute_dtype(prev)
    gw, gb, gx = lin.weight.grad.clone(), lin.bias.grad.clone(), x.grad.clone()
    assert gw.dtype == torch.float32 and gb.dtype == torch.float32
    lin.zero_grad(); x.grad = None
    yr = torch.nn.functional.linear(x, lin.weight, lin.bias)
    if act == "gelu":
        yr = torch.nn.functional.gelu(yr)
    yr.square().sum().backward()
    rel = lambda a, b: ((a - b).norm() / b.norm()).item()
    assert rel(y.float(), yr) < 1e-2 and rel(gw, lin.weight.grad) < 2e-2 and rel(gb, lin.bias.grad) < 2e-2
    assert rel(gx, x.grad) < 2e-2


def test_stream_overlap_is_value_neutral(dev):
    """Side-stream execution (image encoder || detector, 2D || 3D text stream) must not change any value:
    the eval-mode hot path with overlap on equals the one with overlap off, repeatedly (race check)."""
    import bench
    from bridgeqa_amd import fusion_ops
    from bridgeqa_amd.blip_vqa_3d import SyntheticTokenizer
    from bridgeqa_amd.hotpath import ScanQAHotPath
    from bridgeqa_amd.med import BertConfig
    torch.manual_seed(0)
    cfg = BertConfig(num_hidden_layers=3, vocab_size=200, max_position_embeddings=64)
    model = ScanQAHotPath(input_feature_dim=3, blip_kwargs=dict(
        med_config=cfg, image_size=64, tokenizer=SyntheticTokenizer(0, 102, 198, 199))).to(dev).eval()
    g = torch.Generator().manual_seed(1)
    B = 4
    batch = {"point_clouds": bench.synth_batch(B, 5000, 3, 7, dev), "phase": "train",
             "images": torch.randn(B, 1, 3, 64, 64, generator=g).to(dev),
             "question": {"input_ids": torch.randint(5, 190, (B, 9), generator=g).to(dev),
                          "attention_mask": torch.ones(B, 9, dtype=torch.long, device=dev)},
             "answer": {"input_ids": torch.randint(5, 190, (B, 4), generator=g).to(dev),
                        "attention_mask": torch.ones(B, 4, dtype=torch.long, device=dev)}}
    prev_dt = fusion_ops.set_compute_dtype(torch.bfloat16)
    try:
        outs = []
        for flag in (False, True, True, True, False):
            prev = fusion_ops.set_overlap(flag)
            with torch.no_grad():
                dd = model(dict(batch))
            torch.cuda.synchronize()
            fusion_ops.set_overlap(prev)
            outs.append((dd["blip_loss"].float().clone(), dd["fused_feat"].float().clone(),
                         dd["2d_cross_attention"].float().clone()))
        for o in outs[1:]:
            for a, b in zip(outs[0], o):
                assert torch.equal(a, b)
    finally:
        fusion_ops.set_compute_dtype(prev_dt)


def _keep_mask(seed, B, H, Lq, Lk, p, dev):
    """The kernel's stateless dropout hash (csrc/attn.hip drop_keep), restated with 64-bit integer tensors."""
    M = 0xFFFFFFFF
    bh = torch.arange(B * H, device=dev, dtype=torch.int64).view(B, H, 1, 1)
    q = torch.arange(Lq, device=dev, dtype=torch.int64).view(1, 1, Lq, 1)
    k = torch.arange(Lk, device=dev, dtype=torch.int64).view(1, 1, 1, Lk)
    x = (seed & M) ^ ((bh * 0x9E3779B1) & M) ^ ((q * 0x85EBCA77) & M) ^ ((k * 0xC2B2AE3D) & M)
    x = x ^ (x >> 16); x = (x * 0x7feb352d) & M; x = x ^ (x >> 15); x = (x * 0x846ca68b) & M; x = x ^ (x >> 16)
    return x >= int(p * 4294967296.0)


@pytest.mark.parametrize("B,H,Lq,Lk,p", [(2, 3, 20, 1045, 0.0), (2, 12, 20, 276, 0.1), (3, 2, 5, 20, 0.1),
                                         (1, 2, 20, 20, 0.25), (2, 2, 150, 70, 0.1)])
def test_masked_cross_attention_with_dropout_fwd_bwd(dev, B, H, Lq, Lk, p):
    """Lq != Lk, additive key mask (-10000 / -1e9 conventions of med.py) and hash dropout: forward and all three
    gradients vs the fp32 composition that uses the SAME keep mask."""
    from bridgeqa_amd import _ext
    g = torch.Generator().manual_seed(Lq * 7 + Lk)
    q = torch.randn(B, Lq, H, 64, generator=g).to(dev).to(torch.bfloat16)
    k = torch.randn(B, Lk, H, 64, generator=g).to(dev).to(torch.bfloat16)
    v = torch.randn(B, Lk, H, 64, generator=g).to(dev).to(torch.bfloat16)
    go = torch.randn(B, Lq, H, 64, generator=g).to(dev).to(torch.bfloat16)
    valid = torch.ones(B, Lk, device=dev)
    valid[0, Lk // 2:] = 0
    valid[-1, -3:] = 0
    mask = ((1.0 - valid) * (-10000.0 if Lk == 20 else -1e9)).view(B, 1, 1, Lk)
    seed = 12345
    ml2 = _ext.key_mask_log2(mask, B, Lk)
    out, lse = _ext.attn_fwd(q, k, v, 0.125, ml2, p, seed, None)
    dq = torch.empty_like(q); dk = torch.empty_like(k); dv = torch.empty_like(v)
    _ext.attn_bwd(q, k, v, out, lse, go, 0.125, dq, dk, dv, ml2, p, seed, None)
    qf, kf, vf = (t.float().permute(0, 2, 1, 3).detach().requires_grad_(True) for t in (q, k, v))
    s = torch.matmul(qf, kf.transpose(-1, -2)) * 0.125 + mask
    pr = torch.softmax(s, dim=-1)
    if p > 0:
        pr = pr * _keep_mask(seed, B, H, Lq, Lk, p, dev) / (1.0 - p)
    want = torch.matmul(pr, vf)
    want.backward(go.float().permute(0, 2, 1, 3))
    rel = lambda a, b: ((a - b).norm() / b.norm()).item()
    assert rel(out.float().permute(0, 2, 1, 3), want) < 2e-2
    assert rel(dq.float().permute(0, 2, 1, 3), qf.grad) < 3e-2
    assert rel(dk.float().permute(0, 2, 1, 3), kf.grad) < 3e-2
    assert rel(dv.float().permute(0, 2, 1, 3), vf.grad) < 3e-2
    if p > 0:  # the kept fraction is what it should be
        frac = _keep_mask(seed, B, H, Lq, Lk, p, dev).float().mean().item()
        assert abs(frac - (1 - p)) < 0.02


@pytest.mark.parametrize("p", [0.0, 0.1])
def test_paired_narrow_attention_equals_two_launches(dev, p):
    """bq_attn_fwd_pair / bq_attn_bwd_pair (the two cross-attentions of a twin level in one launch per kernel) against the
    single launches, bit for bit: key masks, dropout with per-side seeds, different key counts (1045 / 276)"""
    from bridgeqa_amd import _ext
    B, H, L = 3, 4, 20
    g = torch.Generator().manual_seed(11)
    mk = lambda *s: torch.randn(*s, generator=g).to(dev).to(torch.bfloat16)
    q = mk(2 * B, L, H, 64)
    kvs = [mk(B, 1045, 2, H, 64), mk(B, 276, 2, H, 64)]
    masks = []
    for Lk in (1045, 276):
        m = torch.zeros(B, 1, 1, Lk, device=dev)
        m[1, 0, 0, Lk - 7:] = -10000.0
        masks.append(_ext.key_mask_log2(m, B, Lk))
    go = mk(2 * B, L, H, 64)
    seeds = [123, 456]
    out1, lse1, d1 = torch.empty_like(q), [], []
    for s_ in range(2):
        r = slice(s_ * B, (s_ + 1) * B)
        _, lse = _ext.attn_fwd(q[r], kvs[s_][:, :, 0], kvs[s_][:, :, 1], 0.125, masks[s_], p, seeds[s_], None, out=out1[r])
        lse1.append(lse)
    dq1 = torch.empty_like(q)
    for s_ in range(2):
        r = slice(s_ * B, (s_ + 1) * B)
        dkv = torch.empty_like(kvs[s_])
        _ext.attn_bwd(q[r], kvs[s_][:, :, 0], kvs[s_][:, :, 1], out1[r], lse1[s_], go[r], 0.125, dq1[r], dkv[:, :, 0],
                      dkv[:, :, 1], masks[s_], p, seeds[s_], None)
        d1.append(dkv)
    out2 = torch.empty_like(q)
    sides = [dict(q=q[s_ * B:(s_ + 1) * B], k=kvs[s_][:, :, 0], v=kvs[s_][:, :, 1], out=out2[s_ * B:(s_ + 1) * B],
                  mask_log2=masks[s_], seed=seeds[s_]) for s_ in range(2)]
    assert _ext.attn_pair_ok(sides[0]["q"], sides[0]["k"], sides[1]["q"], sides[1]["k"])
    lse2 = _ext.attn_fwd_pair(sides, 0.125, p, None)
    assert torch.equal(out2, out1) and torch.equal(lse2[0], lse1[0]) and torch.equal(lse2[1], lse1[1])
    dq2, d2 = torch.empty_like(q), [torch.empty_like(kv) for kv in kvs]
    bsides = [dict(sides[s_], lse=lse2[s_], grad_out=go[s_ * B:(s_ + 1) * B], dq=dq2[s_ * B:(s_ + 1) * B],
                   dk=d2[s_][:, :, 0], dv=d2[s_][:, :, 1]) for s_ in range(2)]
    _ext.attn_bwd_pair(bsides, 0.125, p, None)
    assert torch.equal(dq2, dq1) and torch.equal(d2[0], d1[0]) and torch.equal(d2[1], d1[1])


def test_text_attention_routes_to_kernel_and_matches_composition(dev):
    """fusion_ops.attention in bf16 mode with a key mask == the fp32 composition (eval: no dropout)."""
    from bridgeqa_amd import fusion_ops
    g = torch.Generator().manual_seed(3)
    B, H, Lq, Lk = 2, 12, 20, 1045
    q = torch.randn(B, Lq, H, 64, generator=g).to(dev); k = torch.randn(B, Lk, H, 64, generator=g).to(dev)
    v = torch.randn(B, Lk, H, 64, generator=g).to(dev)
    m = torch.zeros(B, 1, 1, Lk, device=dev); m[1, ..., 1000:] = -1e9
    want, _ = fusion_ops.attention(q, k, v, m, 0.125)
    prev = fusion_ops.set_compute_dtype(torch.bfloat16)
    try:
        got, probs = fusion_ops.attention(q, k, v, m, 0.125)
    finally:
        fusion_ops.set_compute_dtype(prev)
    assert probs is None and got.dtype == torch.bfloat16
    assert ((got.float() - want).norm() / want.norm()).item() < 2e-2


@pytest.mark.parametrize("B,H,L", [(3, 12, 5), (2, 4, 33), (1, 2, 64), (2, 3, 130)])
def test_causal_decoder_self_attention_fwd_bwd(dev, B, H, L):
    """decoder self-attention (med.py:771-830 causal AND padding mask) through the kernels in factored form
    (key mask + causal flag) == fp32 composition with the full (B,1,L,L) additive mask, values and gradients."""
    from bridgeqa_amd import fusion_ops as ops
    g = torch.Generator().manual_seed(L)
    qkv = torch.randn(B, L, 3, H, 64, generator=g).to(dev)
    am = torch.ones(B, L, device=dev); am[0, L - 1:] = 0  # one padded key in sample 0
    ids = torch.arange(L, device=dev)
    causal = (ids[None, None, :] <= ids[None, :, None]).float().expand(B, L, L)
    full = (1.0 - causal[:, None] * am[:, None, None, :]) * -10000.0
    key = (1.0 - am[:, None, None, :]) * -10000.0
    go = torch.randn(B, L, H, 64, generator=g).to(dev)
    q32 = qkv.clone().requires_grad_(True)
    want, _ = ops.attention(q32[:, :, 0], q32[:, :, 1], q32[:, :, 2], full, 0.125)
    want.backward(go)
    prev = ops.set_compute_dtype(torch.bfloat16)
    try:
        qb = qkv.to(torch.bfloat16).requires_grad_(True)
        assert ops.packed_kernel_ok(qb, key)
        got = ops.attention_packed(qb, 0.125, 0.0, key, causal=True)
        got.backward(go.to(torch.bfloat16))
    finally:
        ops.set_compute_dtype(prev)
    rel = lambda a, b: ((a.float() - b).norm() / b.norm()).item()
    assert rel(got, want) < 2e-2
    assert rel(qb.grad, q32.grad) < 3e-2


@pytest.mark.parametrize("M,H,p", [(320, 768, 0.0), (320, 768, 0.1), (37, 256, 0.1), (5000, 768, 0.1), (3, 1024, 0.5)])
def test_fused_dropout_add_layernorm_fwd_bwd(dev, M, H, p):
    """csrc/ln.hip vs the reference composition LayerNorm(dropout(x) + residual) with the SAME keep mask."""
    from bridgeqa_amd import _ext
    g = torch.Generator().manual_seed(M)
    x = torch.randn(M, H, generator=g).to(dev).to(torch.bfloat16)
    res = torch.randn(M, H, generator=g).to(dev).to(torch.bfloat16)
    gamma = (torch.rand(H, generator=g) + 0.5).to(dev); beta = (torch.randn(H, generator=g) * 0.1).to(dev)
    dy = torch.randn(M, H, generator=g).to(dev).to(torch.bfloat16)
    seed = 777
    y, _, mean, rstd, _ = _ext.drop_add_ln_fwd(x, res, gamma, beta, 1e-12, p, seed, None)
    dx, dres, dg, db = _ext.drop_add_ln_bwd(x, res, gamma, dy, mean, rstd, 1e-12, p, seed, None)
    # reference with the same hash
    Mk = 0xFFFFFFFF
    r = torch.arange(M, device=dev, dtype=torch.int64).view(M, 1); c = torch.arange(H, device=dev, dtype=torch.int64).view(1, H)
    h = (seed & Mk) ^ ((r * 0x85EBCA77) & Mk) ^ ((c * 0xC2B2AE3D) & Mk)
    h = h ^ (h >> 16); h = (h * 0x7feb352d) & Mk; h = h ^ (h >> 15); h = (h * 0x846ca68b) & Mk; h = h ^ (h >> 16)
    keep = (h >= int(p * 4294967296.0)).float()
    xf, rf = x.float().requires_grad_(True), res.float().requires_grad_(True)
    gf, bf = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    z = xf * keep / (1.0 - p) + rf
    want = torch.nn.functional.layer_norm(z, (H,), gf, bf, 1e-12)
    want.backward(dy.float())
    rel = lambda a, b: ((a.float() - b).norm() / b.norm()).item()
    assert rel(y, want) < 1e-2
    assert rel(dx, xf.grad) < 2e-2 and rel(dres, rf.grad) < 2e-2
    assert rel(dg, gf.grad) < 2e-2 and rel(db, bf.grad) < 2e-2


@pytest.mark.parametrize("M,H,with_res", [(300, 768, True), (300, 768, False), (4105, 768, True), (9, 512, False)])
def test_add_layernorm_with_sum_output_and_plain_form(dev, M, H, with_res):
    """pre-LN form of csrc/ln.hip: (s, LayerNorm(s)) with s = x + residual, gradients arriving at BOTH outputs;
    and the plain LayerNorm(x) form (residual NULL) -- vs the fp32 composition through ops (vit.py:106-109)."""
    from bridgeqa_amd import fusion_ops as ops
    ops.set_compute_dtype(torch.bfloat16)
    try:
        g = torch.Generator().manual_seed(M + H)
        ln = torch.nn.LayerNorm(H, eps=1e-6).to(dev)
        with torch.no_grad():
            ln.weight.copy_(torch.rand(H, generator=g) + 0.5); ln.bias.copy_(torch.randn(H, generator=g) * 0.1)
        x = torch.randn(M, H, generator=g).to(dev).to(torch.bfloat16).requires_grad_(True)
        res = torch.randn(M, H, generator=g).to(dev).to(torch.bfloat16).requires_grad_(True)
        dy = torch.randn(M, H, generator=g).to(dev).to(torch.bfloat16)
        ds = torch.randn(M, H, generator=g).to(dev).to(torch.bfloat16)
        xf, rf = x.detach().float().requires_grad_(True), res.detach().float().requires_grad_(True)
        if with_res:
            s, y = ops.add_layer_norm(x, res, ln)
            assert s.dtype == torch.bfloat16 and y.dtype == torch.bfloat16
            (y.float() * dy.float()).sum().add((s.float() * ds.float()).sum()).backward()
            g_w, g_b = ln.weight.grad.clone(), ln.bias.grad.clone(); ln.zero_grad()
            sf = xf + rf
            yf = torch.nn.functional.layer_norm(sf, (H,), ln.weight, ln.bias, 1e-6)
            ((yf * dy.float()).sum() + (sf * ds.float()).sum()).backward()
        else:
            y = ops.layer_norm(x, ln)
            (y.float() * dy.float()).sum().backward()
            g_w, g_b = ln.weight.grad.clone(), ln.bias.grad.clone(); ln.zero_grad()
            yf = torch.nn.functional.layer_norm(xf, (H,), ln.weight, ln.bias, 1e-6)
            (yf * dy.float()).sum().backward()
        rel = lambda a, b: ((a.float() - b.float()).norm() / b.float().norm()).item()
        assert rel(y, yf) < 1e-2
        assert rel(x.grad, xf.grad) < 2e-2
        if with_res:
            assert rel(s, sf) < 1e-2 and rel(res.grad, rf.grad) < 2e-2 and torch.equal(x.grad, res.grad)
        assert rel(g_w, ln.weight.grad) < 2e-2 and rel(g_b, ln.bias.grad) < 2e-2
    finally:
        ops.set_compute_dtype(torch.float32)


@pytest.mark.parametrize("p_path", [0.0, 0.3])
def test_add_layernorm_backward_from_the_stored_sum(dev, p_path):
    """round 6 (bq_drop_add_ln_bwd_sum): the backward of a dropout-free add + LayerNorm site that wrote its sum re-forms the
    normalised row from that stored bf16 sum instead of from x and residual -- same gradients as the two-operand backward to
    one bf16 rounding of the row (5e-3 rel-L2), the same values for dx and dresidual without stochastic depth, dropped
    samples' dx exactly zero with it"""
    from bridgeqa_amd import fusion_ops as ops
    ops.set_compute_dtype(torch.bfloat16)
    try:
        B, L, H = 8, 1025, 768
        g = torch.Generator().manual_seed(5)
        ln = torch.nn.LayerNorm(H, eps=1e-6).to(dev)
        with torch.no_grad():
            ln.weight.copy_(torch.rand(H, generator=g) + 0.5); ln.bias.copy_(torch.randn(H, generator=g) * 0.1)
        x0 = (torch.randn(B, L, H, generator=g) * 2).to(dev).to(torch.bfloat16)
        r0 = (torch.randn(B, L, H, generator=g) * 3 + 0.5).to(dev).to(torch.bfloat16)
        dy = torch.randn(B, L, H, generator=g).to(dev).to(torch.bfloat16)
        ds = torch.randn(B, L, H, generator=g).to(dev).to(torch.bfloat16)
        res = {}
        for from_sum in (False, True):
            ops._LN_BWD_FROM_SUM[0] = from_sum
            ops._CALL_SEED[0] = 1234          # (the same stochastic-depth draw in both arms)
            x, r = x0.clone().requires_grad_(True), r0.clone().requires_grad_(True)
            ln.zero_grad()
            s, y = ops.add_layer_norm(x, r, ln, drop_path=p_path)
            (y.float() * dy.float()).sum().add((s.float() * ds.float()).sum()).backward()
            res[from_sum] = (s.detach().clone(), y.detach().clone(), x.grad.clone(), r.grad.clone(), ln.weight.grad.clone(),
                             ln.bias.grad.clone(), x.grad.data_ptr() == r.grad.data_ptr())
        a, b = res[False], res[True]
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        rel = lambda u, v: ((u.float() - v.float()).norm() / v.float().norm()).item()
        assert rel(b[2], a[2]) < 5e-3 and rel(b[3], a[3]) < 5e-3 and rel(b[4], a[4]) < 5e-3 and rel(b[5], a[5]) < 5e-3
        if p_path == 0.0:
            assert torch.equal(b[2], b[3])   # (one tensor from the kernel; autograd clones it for the second leaf)
        else:
            dropped = (a[0] == r0).flatten(1).all(1)          # samples whose branch was dropped: sum == residual
            assert dropped.any() and not dropped.all()
            assert torch.equal(b[2][dropped], torch.zeros_like(b[2][dropped])) and (b[2][~dropped] != 0).any()
    finally:
        ops._LN_BWD_FROM_SUM[0] = True
        ops.set_compute_dtype(torch.float32)


def test_add_layernorm_stochastic_depth_drops_whole_samples(dev):
    """vit.py:107-108 x + drop_path(f): with p_path > 0 a sample's branch rows are either all dropped (sum == residual)
    or all scaled by 1 / keep; the backward sends the same scale to dx and an unscaled gradient to the residual."""
    from bridgeqa_amd import _ext
    B, L, H, p = 64, 7, 768, 0.3
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, L, H, generator=g).to(dev).to(torch.bfloat16)
    res = torch.randn(B, L, H, generator=g).to(dev).to(torch.bfloat16)
    gamma, beta = torch.ones(H, device=dev), torch.zeros(H, device=dev)
    y, s, mean, rstd, _ = _ext.drop_add_ln_fwd(x, res, gamma, beta, 1e-6, 0.0, 99, None, True, p, L)
    d = (s.float() - res.float())
    dropped = d.abs().amax(dim=(1, 2)) == 0
    assert 5 <= int(dropped.sum()) <= 35  # ~ Binomial(64, 0.3)
    kept = ~dropped
    want = x.float() / (1 - p) + res.float()
    assert ((s.float() - want)[kept].abs().max() <= 0.04 * want[kept].abs().max()).item()
    dy = torch.zeros_like(x); dsum = torch.ones_like(x)
    dx, dres, _, _ = _ext.drop_add_ln_bwd(x, res, gamma, dy, mean, rstd, 1e-6, 0.0, 99, None, dsum, p, L)
    assert torch.all(dres.float() == 1)
    assert torch.all(dx[dropped].float() == 0) and torch.allclose(dx[kept].float(), torch.tensor(1 / (1 - p), device=dev), rtol=1e-2)
    # in LayerNorm terms: y is the norm of s
    ref = torch.nn.functional.layer_norm(s.float(), (H,), gamma, beta, 1e-6)
    assert ((y.float() - ref).norm() / ref.norm()).item() < 1e-2


def test_twin_encoder_and_decoder_bf16_fused_path_vs_fp32(dev):
    """Twin encoder + LM decoder through every fused piece (multi-linear QKV / KV, packed masked attention, fused
    dropout+add+LayerNorm, fp32-master linear) in eval mode vs the fp32 reference composition; then a train-mode
    backward through the fused path must give finite gradients of plausible size for every used parameter."""
    from bridgeqa_amd import fusion_ops, med
    cfg = med.BertConfig(hidden_size=256, num_attention_heads=4, intermediate_size=512, num_hidden_layers=2,
                         vocab_size=200, max_position_embeddings=64, encoder_width=256)
    torch.manual_seed(0)
    twin = med.BertModelTwin(config=cfg, add_pooling_layer=False).to(dev).eval()
    dec = med.BertLMHeadModel(config=cfg).to(dev).eval()
    g = torch.Generator().manual_seed(1)
    B, L, P, O = 3, 9, 70, 11
    ids = torch.randint(5, 190, (B, L), generator=g).to(dev)
    am = torch.ones(B, L, dtype=torch.long, device=dev); am[1, 6:] = 0
    img = torch.randn(B, P, 256, generator=g).to(dev); obj = torch.randn(B, O, 256, generator=g).to(dev)
    om = torch.ones(B, O, dtype=torch.long, device=dev); om[0, 7:] = 0

    def run():
        r = twin(ids, attention_mask=am, encoder_hidden_states=img,
                 encoder_attention_mask=torch.ones(B, P, dtype=torch.long, device=dev),
                 encoder_hidden_states_twin=obj, encoder_attention_mask_twin=om, output_attentions="last")
        h2d, h3d = r.last_hidden_state
        aid = ids[:, :5].clone(); aid[:, 0] = 198
        d = dec(aid, attention_mask=torch.ones_like(aid), encoder_hidden_states=h2d, encoder_attention_mask=am,
                labels=aid, reduction="none")
        return h2d.float(), h3d.float(), d.loss.float(), r.cross_attentions[-1][0].float()
    with torch.no_grad():
        want = run()
        prev = fusion_ops.set_compute_dtype(torch.bfloat16)
        try:
            got = run()
        finally:
            fusion_ops.set_compute_dtype(prev)
    for a, b, tol in zip(got, want, (3e-2, 3e-2, 2e-2, 3e-2)):
        assert ((a - b).norm() / b.norm()).item() < tol
    twin.train(); dec.train()
    prev = fusion_ops.set_compute_dtype(torch.bfloat16)
    try:
        h2d, h3d, loss, _ = run()
        (loss.sum() + h3d.square().mean()).backward()
    finally:
        fusion_ops.set_compute_dtype(prev)
    for name, p in list(twin.named_parameters()) + list(dec.named_parameters()):
        if "LayerNorms.0" in name or "pooler" in name:
            continue
        assert p.grad is not None and torch.isfinite(p.grad).all(), name
    assert twin.encoder.layer_twin[0].crossattention.self.value.weight.grad.abs().sum() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("M,N", [(0, 256), (1, 256), (31, 768), (32, 768), (33, 772), (560, 768), (577, 2304), (1024, 3072), (1500, 768), (2049, 260),
                                 (9232, 3072), (16 * 1025, 768)])
def test_colsum_single_launch_matches_fp64_and_is_reproducible(dev, M, N):
    """bias gradients: f32 column sums of a bf16 matrix, one launch, chunk partials folded by the last workgroup."""
    from bridgeqa_amd import _ext
    torch.manual_seed(M * 7 + N)
    g = torch.randn(M, N, device=dev).to(torch.bfloat16)
    ref = g.double().sum(0)
    outs = [_ext.colsum(g) for _ in range(3)]
    assert outs[0].shape == (N,) and outs[0].dtype == torch.float32
    tol = 2e-6 * max(M, 1) ** 0.5 * 4 + 1e-6  # f32 accumulation of O(1) terms
    assert (outs[0].double() - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])  # fixed summation order
    # many launches in flight on two streams: the rotating completion counters must not collide
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    res = []
    for i in range(20):
        with torch.cuda.stream(s1 if i % 2 else s2):
            res.append(_ext.colsum(g))
    torch.cuda.synchronize()
    assert all(torch.equal(r, outs[0]) for r in res)


def test_deferred_grouped_weight_gradients_are_value_neutral(dev):
    """fusion_ops.begin/flush_deferred_wgrad: dW / db parked during the backward and produced by one grouped launch
    == computed node by node (same kernels and tile classes: dW bit for bit; db to fp32 atomics' reordering)."""
    from bridgeqa_amd import fusion_ops as ops
    prev_dt = ops.set_compute_dtype(torch.bfloat16)
    try:
        torch.manual_seed(0)
        lin = torch.nn.Linear(768, 3072).to(dev)
        trio = [torch.nn.Linear(768, 768).to(dev) for _ in range(3)]
        fc1, fc2 = torch.nn.Linear(768, 3072).to(dev), torch.nn.Linear(3072, 768).to(dev)
        x = torch.randn(16, 20, 768, device=dev).to(torch.bfloat16)
        xl = torch.randn(4, 600, 768, device=dev).to(torch.bfloat16)
        mods = [lin] + trio + [fc1, fc2]

        def run(defer):
            for m in mods:
                m.zero_grad(set_to_none=True)
            xa, xb = x.clone().requires_grad_(True), xl.clone().requires_grad_(True)
            y = ops.linear(xa, lin.weight, lin.bias, act="gelu").float().square().mean()
            z = ops.multi_linear(xa, trio).float().square().mean()
            u = ops.mlp(xb, fc1, fc2).float().square().mean() + ops.mlp(xa, fc1, fc2).float().square().mean()
            if defer:
                ops.begin_deferred_wgrad()
            try:
                (y + z + u).backward()
            finally:
                ops.flush_deferred_wgrad()
            torch.cuda.synchronize()
            return [p.grad.clone() for m in mods for p in m.parameters()] + [xa.grad.clone(), xb.grad.clone()]
        a, b = run(False), run(True)
        for u, v in zip(a, b):
            assert torch.isfinite(u).all() and u.abs().max().item() > 0
            assert ((u - v).abs().max() / (u.abs().max() + 1e-30)).item() < 1e-5
    finally:
        ops.set_compute_dtype(prev_dt)


def test_deferred_weight_gradients_with_a_launch_plan_that_moves_problems(dev):
    """22 problems of 12 tiles = 264 tiles: 2 rounds on 256 CUs; the plan sends one to the 64 x 64-tile kernel (252 tiles,
    one round).  Whatever the plan, every dW / db must equal the inline computation."""
    from bridgeqa_amd import fusion_ops as ops
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    n = max(2, cus // 12 + 1)
    g = torch.Generator().manual_seed(5)
    lins = [torch.nn.Linear(768, 1024).to(dev) for _ in range(n)]
    xs = [torch.randn(1024, 768, generator=g).to(dev).to(torch.bfloat16) for _ in range(n)]
    prev = ops.set_compute_dtype(torch.bfloat16)
    try:
        def run(deferred):
            for l in lins:
                l.zero_grad(set_to_none=True)
            if deferred:
                ops.begin_deferred_wgrad()
            for l, x in zip(lins, xs):
                ops.linear(x, l.weight, l.bias).float().square().sum().backward()
            if deferred:
                ops.flush_deferred_wgrad()
            return [(l.weight.grad.clone(), l.bias.grad.clone()) for l in lins]
        inline, parked = run(False), run(True)
        groups, moved = ops.plan_big_launches([12] * n, cus)
        assert len(moved) >= 1 or cus != 256          # (on a 256-CU part the plan does move a problem)
        for (dw0, db0), (dw1, db1) in zip(inline, parked):
            assert (dw1 - dw0).norm() <= 1e-5 * dw0.norm() and (db1 - db0).norm() <= 1e-5 * db0.norm()
    finally:
        ops.set_compute_dtype(prev)


def test_deferred_batched_weight_gradients_match_inline(dev):
    """fusion_ops.begin/flush_deferred_wgrad: parked dW / db of same-shape linears computed as one batched GEMM + one
    reduction == computed per layer inside the backward (fp32 accumulation either way; kernels differ -> 1e-3)."""
    from bridgeqa_amd import fusion_ops as ops
    prev_dt = ops.set_compute_dtype(torch.bfloat16)
    try:
        torch.manual_seed(1)
        stack = [torch.nn.Linear(768, 768).to(dev) for _ in range(5)]
        trios = [[torch.nn.Linear(768, 768).to(dev) for _ in range(3)] for _ in range(3)]
        lone = torch.nn.Linear(768, 256).to(dev)
        x = torch.randn(16, 20, 768, device=dev).to(torch.bfloat16)
        mods = stack + [m for t in trios for m in t] + [lone]

        def run(defer):
            for m in mods:
                m.zero_grad(set_to_none=True)
            h = x.clone().requires_grad_(True)
            y = h
            if defer:
                ops.begin_deferred_wgrad()
            for l, t in zip(stack, trios + [None, None]):
                y = ops.linear(y, l.weight, l.bias, act="gelu")
                if t is not None:
                    y = y + ops.multi_linear(y, t).sum(-2)
            (ops.linear(y, lone.weight, lone.bias).float().square().mean()).backward()
            if defer:
                assert all(m.weight.grad is None for m in mods)  # parked, not yet computed
                ops.flush_deferred_wgrad()
            torch.cuda.synchronize()
            return [p.grad.clone() for m in mods for p in m.parameters()] + [h.grad.clone()]
        a, b = run(False), run(True)
        rel = lambda u, v: ((u.float() - v.float()).norm() / (v.float().norm() + 1e-20)).item()
        assert max(rel(u, v) for u, v in zip(b, a)) < 1e-3
    finally:
        ops.set_compute_dtype(prev_dt)


def test_shadow_weights_follow_the_optimizer(dev):
    """bf16 operand copies of fp32 master weights (single and concatenated QKV form): refreshed by ONE multi-tensor
    cast after an in-place update (refresh_shadows), and by the lazy version check when nobody refreshed them."""
    from bridgeqa_amd import fusion_ops as ops
    prev_dt = ops.set_compute_dtype(torch.bfloat16)
    ops._SHADOW.clear(); ops._CAT_CACHE.clear(); ops._PADDED.clear()  # (operand copies other tests' modules registered)
    try:
        torch.manual_seed(2)
        lin = torch.nn.Linear(256, 256).to(dev)
        trio = [torch.nn.Linear(256, 256).to(dev) for _ in range(3)]
        x = torch.randn(4, 7, 256, device=dev).to(torch.bfloat16)

        def outputs():
            with torch.no_grad():
                return ops.linear(x, lin.weight, lin.bias).float(), ops.multi_linear(x, trio).float()

        def reference():
            with torch.no_grad():
                f = lambda m: torch.nn.functional.linear(x, m.weight.to(torch.bfloat16), m.bias.to(torch.bfloat16)).float()
                return f(lin), torch.stack([f(m) for m in trio], dim=-2)
        for mode in ("first use", "refresh", "lazy"):
            if mode != "first use":
                with torch.no_grad():
                    for m in [lin] + trio:
                        m.weight.add_(torch.randn_like(m.weight) * 0.1)
                        m.bias.add_(0.5)
            if mode == "refresh":
                # 4 weights + the 3 concatenated biases, one multi-tensor cast (the single linear's bias has no operand
                # copy: the GEMM epilogue adds the fp32 master bias itself)
                assert ops.refresh_shadows() == 7
                assert ops.refresh_shadows() == 0
            a, b = outputs()
            ra, rb = reference()
            assert torch.allclose(a, ra, atol=2e-2, rtol=2e-2) and torch.allclose(b, rb, atol=2e-2, rtol=2e-2), mode
    finally:
        ops.set_compute_dtype(prev_dt)


@pytest.mark.parametrize("capturable", [False, True])
def test_fused_optimizer_updates_reach_the_bf16_operands(dev, capturable):
    """torch's fused multi-tensor AdamW updates parameters WITHOUT bumping their version counters: the bf16 operand
    copies must still follow (global optimizer post-step hook -> refresh_shadows), for single and fused-QKV linears."""
    from bridgeqa_amd import fusion_ops as ops
    prev_dt = ops.set_compute_dtype(torch.bfloat16)
    try:
        torch.manual_seed(3)
        lin = torch.nn.Linear(256, 256).to(dev)
        trio = [torch.nn.Linear(256, 256).to(dev) for _ in range(3)]
        params = list(lin.parameters()) + [p for m in trio for p in m.parameters()]
        opt = torch.optim.AdamW(params, lr=5e-2, fused=True, capturable=capturable)
        x = torch.randn(4, 7, 256, device=dev).to(torch.bfloat16)
        for _ in range(2):
            y = ops.linear(x, lin.weight, lin.bias).float().square().mean() + ops.multi_linear(x, trio).float().square().mean()
            opt.zero_grad(set_to_none=True)
            y.backward()
            before = lin.weight.detach().clone()
            opt.step()
            assert not torch.equal(before, lin.weight.detach())
            with torch.no_grad():
                f = lambda m: torch.nn.functional.linear(x, m.weight.to(torch.bfloat16), m.bias.to(torch.bfloat16)).float()
                got1, got3 = ops.linear(x, lin.weight, lin.bias).float(), ops.multi_linear(x, trio).float()
                assert torch.allclose(got1, f(lin), atol=2e-2, rtol=2e-2)
                assert torch.allclose(got3, torch.stack([f(m) for m in trio], dim=-2), atol=2e-2, rtol=2e-2)
                stale = torch.nn.functional.linear(x, before.to(torch.bfloat16), lin.bias.to(torch.bfloat16)).float()
                assert not torch.allclose(got1, stale, atol=1e-3, rtol=1e-3)  # lr is large enough to tell them apart
    finally:
        ops.set_compute_dtype(prev_dt)


def test_fused_adamw_matches_torch_adamw_and_writes_the_shadows(dev):
    """optim.FusedAdamW (csrc/adamw.hip) == torch.optim.AdamW over several steps (odd sizes, two parameter groups with
    their own lr / weight decay, a parameter without gradient), and the bf16 shadows it writes are the updated weights."""
    from bridgeqa_amd import fusion_ops as ops
    from bridgeqa_amd.optim import FusedAdamW
    prev_dt = ops.set_compute_dtype(torch.bfloat16)
    try:
        torch.manual_seed(4)
        shapes = [(768, 768), (3072,), (37, 5), (1,), (256, 259)]
        mine = [torch.nn.Parameter(torch.randn(*s, device=dev)) for s in shapes] + [torch.nn.Parameter(torch.randn(8, device=dev))]
        ref = [torch.nn.Parameter(p.detach().clone()) for p in mine]
        lin_w, lin_b = mine[0], mine[1][:768].detach()  # give the first weight a registered shadow
        x = torch.randn(4, 768, device=dev).to(torch.bfloat16)
        ops.linear(x, lin_w, None)
        assert ops.shadow_of(lin_w) is not None
        groups = lambda ps: [dict(params=ps[:3], lr=3e-3, weight_decay=1e-2), dict(params=ps[3:], lr=1e-2, weight_decay=0.0)]
        o1 = FusedAdamW(groups(mine), betas=(0.9, 0.999), eps=1e-8)
        o2 = torch.optim.AdamW(groups(ref), betas=(0.9, 0.999), eps=1e-8)
        for step in range(5):
            for a, b in zip(mine[:-1], ref[:-1]):  # the last parameter never gets a gradient
                g = torch.randn_like(a)
                a.grad, b.grad = g.clone(), g.clone()
            o1.step(); o2.step()
            for a, b in zip(mine, ref):
                assert torch.allclose(a, b, rtol=2e-6, atol=2e-7), (step, a.shape, (a - b).abs().max().item())
            assert torch.equal(ops.shadow_of(lin_w), lin_w.detach().to(torch.bfloat16))
            got = ops.linear(x, lin_w, None).float()
            want = torch.nn.functional.linear(x, lin_w.detach().to(torch.bfloat16)).float()
            assert torch.allclose(got, want, atol=2e-2, rtol=2e-2)
        assert torch.equal(mine[-1], ref[-1])
        # gradient value clipping inside the kernel == clip_grad_value_ then AdamW (lib/solver.py:407-409)
        mc = [torch.nn.Parameter(torch.randn(300, 7, device=dev)), torch.nn.Parameter(torch.randn(1001, device=dev))]
        rc = [torch.nn.Parameter(p.detach().clone()) for p in mc]
        o3, o4 = FusedAdamW(mc, lr=1e-2, grad_clip_value=1.0), torch.optim.AdamW(rc, lr=1e-2)
        for step in range(3):
            for a, b in zip(mc, rc):
                g = torch.randn_like(a) * 3
                a.grad, b.grad = g.clone(), g.clone()
            torch.nn.utils.clip_grad_value_(rc, 1.0)
            o3.step(); o4.step()
            for a, b in zip(mc, rc):
                assert torch.allclose(a, b, rtol=2e-6, atol=2e-7)
            assert mc[0].grad.abs().max().item() > 1.0   # the stored gradients stay unclamped
    finally:
        ops.set_compute_dtype(prev_dt)


def test_gelu_kernel_matches_torch_exact_gelu(dev):
    from bridgeqa_amd import _ext
    x = (torch.randn(3, 1025, 3072, device=dev) * 2).to(torch.bfloat16)
    got = _ext.gelu_fwd(x)
    want = torch.nn.functional.gelu(x)
    assert got.dtype == torch.bfloat16 and (got.float() - want.float()).abs().max().item() <= 2 ** -7 * max(1.0, want.float().abs().max().item()) * 0.01 + 1e-2
    assert ((got.float() - want.float()).norm() / want.float().norm()).item() < 2e-3


def test_fused_adamw_subset_steps_equal_one_step(dev):
    """FusedAdamW.step(subset=...): a training step's update issued as two launches over disjoint parameter subsets
    (pipeline.PhasedTrainStep steps the fusion parameters early, the rest at the end) == torch.optim.AdamW's one step;
    the membership is fixed by the first call of a subset; the update counter advances once."""
    from bridgeqa_amd.optim import FusedAdamW
    torch.manual_seed(7)
    shapes = [(300, 7), (1001,), (64, 64), (5,)]
    mine = [torch.nn.Parameter(torch.randn(*s, device=dev)) for s in shapes]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in mine]
    o1, o2 = FusedAdamW(mine, lr=1e-2, weight_decay=1e-2), torch.optim.AdamW(ref, lr=1e-2, weight_decay=1e-2)
    for step in range(4):
        grads = [torch.randn_like(p) for p in mine]
        for b, g in zip(ref, grads):
            b.grad = g.clone()
        for p in mine:
            p.grad = None
        for p, g in zip(mine, grads):   # every gradient is there at both calls: the membership decides
            p.grad = g.clone()
        o1.step(subset="early", params=mine[:2] if step == 0 else None, advance=True)
        o1.step(subset="late", params=mine[2:] if step == 0 else None, advance=False)
        o2.step()
        for a, b in zip(mine, ref):
            assert torch.allclose(a, b, rtol=2e-6, atol=2e-7), (step, a.shape, (a - b).abs().max().item())
    assert float(o1.state[mine[0]]["step"]) == 4.0


@pytest.mark.parametrize("B,H,Lq,Lk,masked,causal", [(2, 4, 20, 1045, True, False), (1, 2, 20, 276, False, False),
                                                      (2, 3, 6, 6, True, True), (1, 2, 77, 77, False, False),
                                                      (2, 2, 1, 300, True, False)])
def test_attention_probabilities_rebuilt_from_the_lse(dev, B, H, Lq, Lk, masked, causal):
    """bq_attn_probs: the softmax map of a fused forward, rebuilt from Q, K and its LSE (what output_attentions returns
    on the kernel path, reference med.py:202,223) == softmax(q k^T * scale + mask) of the composition; rows sum to 1"""
    from bridgeqa_amd import _ext
    g = torch.Generator().manual_seed(Lk)
    q = torch.randn(B, Lq, H, 64, generator=g).to(dev).to(torch.bfloat16)
    kv = torch.randn(B, Lk, 2, H, 64, generator=g).to(dev).to(torch.bfloat16)
    mask = None
    if masked:
        keep = torch.ones(B, Lk)
        keep[0, Lk - min(Lk - 1, 3):] = 0
        mask = ((1.0 - keep) * -10000.0).view(B, 1, 1, Lk).to(dev)
    ml = _ext.key_mask_log2(mask, B, Lk) if mask is not None else None
    out, lse = _ext.attn_fwd(q, kv[:, :, 0], kv[:, :, 1], 0.125, ml, 0.0, 0, None, causal)
    P = _ext.attn_probs(q, kv[:, :, 0], lse, 0.125, ml, causal=causal)
    s = torch.einsum("bqhd,bkhd->bhqk", q.float(), kv[:, :, 0].float()) * 0.125
    if mask is not None:
        s = s + mask
    if causal:
        s = s + torch.triu(torch.full((Lq, Lk), -1e9, device=dev), 1)
    want = torch.softmax(s, -1)
    assert P.shape == want.shape and torch.isfinite(P).all()
    assert (P - want).abs().max().item() < 2e-5 + 2e-3 * want.max().item()
    assert (P.sum(-1) - 1).abs().max().item() < 1e-3
    # and it is the map the fused forward used: P V == the kernel's context (bf16 tolerance)
    ctx = torch.einsum("bhqk,bkhd->bqhd", P, kv[:, :, 1].float())
    assert ((ctx - out.float()).norm() / out.float().norm()).item() < 1e-2

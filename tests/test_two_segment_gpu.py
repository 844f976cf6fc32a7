"""Two-segment attention kernels (bq_attn_fwd2 / bq_attn_bwd2) against the single-segment kernels on the concatenated
keys / values (same arithmetic, different tiling: rel-L2 1e-2 on bf16 outputs)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,H,Lq,L1,L2", [(2, 3, 20, 1025, 20), (1, 2, 5, 70, 3), (2, 12, 20, 256, 20), (1, 1, 32, 64, 64)])
def test_two_segment_attention_matches_concatenated(dev, B, H, Lq, L1, L2):
    from bridgeqa_amd import _ext
    torch.manual_seed(0)
    q = torch.randn(B, Lq, H, 64, device=dev).to(torch.bfloat16)
    big = torch.randn(B, L1, 3, 2, H, 64, device=dev).to(torch.bfloat16)
    kv1 = big[:, :, 1]                                    # strided view, as a hoisted projection's layer slot is
    kv2 = torch.randn(B, L2, 2, H, 64, device=dev).to(torch.bfloat16)
    go = torch.randn(B, Lq, H, 64, device=dev).to(torch.bfloat16)
    mask = torch.zeros(B, 1, 1, L1 + L2, device=dev)
    mask[0, 0, 0, 1] = -10000.0
    mask[-1, 0, 0, L1 + L2 - 1] = -10000.0
    mask[-1, 0, 0, L1 - 1] = -10000.0
    scale = 0.125
    out2, lse2 = _ext.attn_fwd2(q, kv1[:, :, 0], kv1[:, :, 1], kv2[:, :, 0], kv2[:, :, 1], scale,
                                _ext.key_mask_log2_two(mask, B, L1, L2))
    cat = torch.cat((kv1, kv2), dim=1).contiguous()
    out1, lse1 = _ext.attn_fwd(q, cat[:, :, 0], cat[:, :, 1], scale, _ext.key_mask_log2(mask, B, L1 + L2))
    rel = lambda a, b: ((a.float() - b.float()).norm() / b.float().norm()).item()
    assert rel(out2, out1) < 1e-2
    assert (lse2 - lse1).abs().max().item() < 2e-3
    # backward: dK/dV of segment 1 written through a strided view (the hoisted projection's gradient buffer)
    gbig = torch.zeros_like(big)
    dkv1 = gbig[:, :, 1]
    dq2, dkv2 = torch.empty_like(q), torch.empty_like(kv2)
    _ext.attn_bwd2(q, kv1[:, :, 0], kv1[:, :, 1], kv2[:, :, 0], kv2[:, :, 1], out2, lse2, go, scale, dq2, dkv1[:, :, 0],
                   dkv1[:, :, 1], dkv2[:, :, 0], dkv2[:, :, 1], _ext.key_mask_log2_two(mask, B, L1, L2))
    dq1, dcat = torch.empty_like(q), torch.empty_like(cat)
    _ext.attn_bwd(q, cat[:, :, 0], cat[:, :, 1], out1, lse1, go, scale, dq1, dcat[:, :, 0], dcat[:, :, 1],
                  _ext.key_mask_log2(mask, B, L1 + L2))
    assert rel(dq2, dq1) < 1e-2
    assert rel(dkv1, dcat[:, :L1]) < 1e-2 and rel(dkv2, dcat[:, L1:]) < 1e-2
    assert gbig[:, :, 0].abs().max().item() == 0 and gbig[:, :, 2].abs().max().item() == 0  # neighbours untouched


def test_two_segment_attention_dropout_is_consistent_between_forward_and_backward(dev):
    """with dropout the keep pattern is a hash of the PADDED key index, so no bitwise twin exists: check the analytic
    gradient against a finite difference of the forward along one direction of q"""
    from bridgeqa_amd import _ext
    torch.manual_seed(1)
    B, H, Lq, L1, L2 = 1, 2, 7, 130, 9
    q = torch.randn(B, Lq, H, 64, device=dev).to(torch.bfloat16)
    kv1 = torch.randn(B, L1, 2, H, 64, device=dev).to(torch.bfloat16)
    kv2 = torch.randn(B, L2, 2, H, 64, device=dev).to(torch.bfloat16)
    go = torch.randn(B, Lq, H, 64, device=dev).to(torch.bfloat16)
    f = lambda qq: _ext.attn_fwd2(qq, kv1[:, :, 0], kv1[:, :, 1], kv2[:, :, 0], kv2[:, :, 1], 0.125, None, 0.3, 77)
    out, lse = f(q)
    dq, d1, d2 = torch.empty_like(q), torch.empty_like(kv1), torch.empty_like(kv2)
    _ext.attn_bwd2(q, kv1[:, :, 0], kv1[:, :, 1], kv2[:, :, 0], kv2[:, :, 1], out, lse, go, 0.125, dq, d1[:, :, 0],
                   d1[:, :, 1], d2[:, :, 0], d2[:, :, 1], None, 0.3, 77)
    d = torch.randn_like(q.float())
    eps = 0.25
    fd = ((f((q.float() + eps * d).to(torch.bfloat16))[0].float() - f((q.float() - eps * d).to(torch.bfloat16))[0].float())
          * go.float()).sum().item() / (2 * eps)
    an = (dq.float() * d).sum().item()
    assert abs(fd - an) < 0.15 * max(abs(an), 1.0), (fd, an)
    assert torch.isfinite(d1.float()).all() and torch.isfinite(d2.float()).all()


def test_twin_encoder_hoisted_two_segment_wiring_on_gpu(dev):
    """BertModelTwin with med._TWO_SEGMENT (hoisted K/V GEMM + two-segment kernels + in-place gradient sink) against
    the default wiring (per-layer cat + single-segment kernels): both streams' outputs, the gradients of the image /
    object tokens and of every parameter."""
    from bridgeqa_amd import fusion_ops as ops, med
    cfg = med.BertConfig(hidden_size=256, num_attention_heads=4, intermediate_size=512, num_hidden_layers=3,
                         vocab_size=200, max_position_embeddings=64, encoder_width=256)
    torch.manual_seed(0)
    twin = med.BertModelTwin(config=cfg, add_pooling_layer=False).to(dev).eval()
    B, L, P, O = 3, 9, 150, 11
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(5, 190, (B, L), generator=g).to(dev)
    am = torch.ones(B, L, dtype=torch.long, device=dev); am[1, 6:] = 0
    img0 = torch.randn(B, P, 256, generator=g).to(dev); obj0 = torch.randn(B, O, 256, generator=g).to(dev)
    om = torch.ones(B, O, dtype=torch.long, device=dev); om[0, 7:] = 0
    prev = ops.set_compute_dtype(torch.bfloat16)
    flag = med._TWO_SEGMENT
    try:
        def run(hoisted):
            med._TWO_SEGMENT = hoisted
            twin.zero_grad()
            img, obj = img0.clone().requires_grad_(True), obj0.clone().requires_grad_(True)
            r = twin(ids, attention_mask=am, encoder_hidden_states=img,
                     encoder_attention_mask=torch.ones(B, P, dtype=torch.long, device=dev),
                     encoder_hidden_states_twin=obj, encoder_attention_mask_twin=om, return_dict=True, output_attentions="last")  # as BLIP_VQA3D calls it
            h2d, h3d = r.last_hidden_state
            assert len(r.cross_attentions) == 1
            # (a well-conditioned functional: random projections.  sum(h^2) of LayerNorm outputs is nearly invariant to
            # everything upstream: its gradients are ~1e-5 and made of rounding noise)
            gw = torch.Generator().manual_seed(9)
            w2 = torch.randn(h2d.shape, generator=gw).to(dev)
            w3 = torch.randn(h3d.shape, generator=gw).to(dev)
            ((h2d.float() * w2).sum() + (h3d.float() * w3).sum()
             + r.cross_attentions[-1][0].float().square().sum()).backward()
            grads = {n: p.grad.float().clone() for n, p in twin.named_parameters() if p.grad is not None}
            return h2d.detach().float(), h3d.detach().float(), img.grad.float(), obj.grad.float(), grads
        over = ops.set_overlap(False)  # the hoisted projection is single-stream wiring (med.py: not overlap_enabled)
        try:
            a = run(True)
            b = run(False)
        finally:
            ops.set_overlap(over)
    finally:
        med._TWO_SEGMENT = flag
        ops.set_compute_dtype(prev)
    rel = lambda x, y: ((x - y).norm() / (y.norm() + 1e-12)).item()
    for x, y in zip(a[:4], b[:4]):
        assert rel(x, y) < 3e-2, rel(x, y)
    assert a[4].keys() == b[4].keys()
    # (key biases excluded: softmax is invariant to a constant added to every score of a query, so their gradient is
    # exactly zero in exact arithmetic and rounding noise in floating point)
    live = [k for k in a[4] if b[4][k].norm().item() > 1e-2 and not k.endswith("key.bias")]
    assert len(live) > 60
    worst = max((rel(a[4][k], b[4][k]), k) for k in live)
    assert worst[0] < 6e-2, worst


def test_phased_pipeline_trains_with_two_segment_path(dev):
    """tests/test_pipeline_gpu.py::test_phased_step_graph_replay_trains with med._TWO_SEGMENT on: the captured fusion
    phase (hoisted projection, gradient sink allocated inside the capture, FusedAdamW writing the re-registered
    concatenated shadows) must train"""
    import bench
    from bridgeqa_amd import fusion_ops as ops, med
    from bridgeqa_amd.optim import FusedAdamW
    from bridgeqa_amd.pipeline import PhasedTrainStep
    from test_pipeline_gpu import _batch, _small_model
    prev = ops.set_compute_dtype(torch.bfloat16)
    flag = med._TWO_SEGMENT
    med._TWO_SEGMENT = True
    prev_overlap = ops.set_overlap(False)   # the hoisted path is taken on the single-stream fusion graph only
    built = []
    init = ops.HoistedKV.__init__
    ops.HoistedKV.__init__ = lambda self, *a, **k: (built.append(1), init(self, *a, **k))[1]
    try:
        model = _small_model(dev)
        batch = _batch(dev)
        opt = FusedAdamW(model.parameters(), lr=1e-3)
        pipe = PhasedTrainStep(model, batch, bench.det_loss, bench.fusion_loss, opt, use_graphs=True).capture(warmup=3)
        assert len(built) >= 2   # one hoisted projection per stream and forward: the two-segment path really ran
        w = model.blip_model.text_encoder.encoder.layer[0].crossattention.self.value.weight
        w0 = w.detach().clone()
        losses = []
        for _ in range(16):
            l = pipe.step()
            pipe.wait()
            torch.cuda.synchronize()
            losses.append(l.item())
        assert all(x == x and abs(x) < 1e6 for x in losses), losses
        assert not torch.equal(w0, w.detach())
        assert min(losses[-3:]) < 0.9 * losses[0], losses
    finally:
        ops.HoistedKV.__init__ = init
        ops.set_overlap(prev_overlap)
        med._TWO_SEGMENT = flag
        ops.set_compute_dtype(prev)

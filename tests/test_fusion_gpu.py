"""The fusion half (ViT / twin MED encoder / LM decoder / BLIP_VQA3D) ON THE GPU against golden vectors produced by
the reference's own Python (oracle/gen_golden_fusion.py): rows a14-a17 of SURVEY.md §8a.

  * fp32 on cuda: the mirrors' arithmetic on the device (1e-3, the same fixtures as tests/test_fusion_cpu.py).
  * bf16 through the HIP kernels -- MFMA GEMMs with fused epilogues (csrc/gemm.hip), fused attention (csrc/attn.hip),
    fused add + LayerNorm (csrc/ln.hip) -- on the kernel-shaped fixtures (head dim 64, widths % 256): outputs AND
    gradients against the reference's fp32 numbers.  Tolerances ("bf16-in/fp32-acc; vs fp32
    oracle rel-L2 <= 2e-2 on final tokens", "loss fp32, tol 1e-2 rel"): rel-L2 <= 2e-2 on outputs, 1e-2 relative on
    losses, rel-L2 <= 5e-2 on parameter / input gradients (two bf16 roundings per layer on the way back)."""
import numpy as np
import pytest
import torch

from golden_util import subsample
from test_fusion_cpu import keys_of, run_blip, run_twin_and_decoder, run_vit

pytestmark = pytest.mark.gpu


def rel_l2(a, g):
    a = subsample(a.detach().float().cpu().numpy()).astype(np.float64).reshape(-1)
    g = np.asarray(g, dtype=np.float64).reshape(-1)
    assert a.shape == g.shape, (a.shape, g.shape)
    assert np.isfinite(a).all()
    return float(np.linalg.norm(a - g) / (np.linalg.norm(g) + 1e-30))


def grad_of(module, key):
    """fixture key grad_<name with . -> _> -> the parameter's gradient"""
    for n, p in module.named_parameters():
        if n.replace(".", "_") == key:
            return p.grad
    raise KeyError(key)


@pytest.fixture()
def bf16():
    from bridgeqa_amd import fusion_ops as ops
    prev = ops.set_compute_dtype(torch.bfloat16)
    yield ops
    ops.set_compute_dtype(prev)


def test_fp32_on_device_vs_reference_golden(golden, dev):
    run_vit(golden("fusion_vit.npz"), dev, 1e-3, 1e-4)
    run_twin_and_decoder(golden("fusion_med.npz"), dev, 1e-3, 1e-4)
    run_blip(golden("fusion_blip.npz"), dev, 1e-3, 1e-4)


def test_vit_bf16_hip_path_vs_reference_golden(golden, dev, bf16):
    from bridgeqa_amd import vit
    g = golden("fusion_vit_k.npz")
    m = vit.VisionTransformer(img_size=64, patch_size=16, embed_dim=256, depth=2, num_heads=4, drop_path_rate=0.1)
    assert keys_of(m, "visual_encoder.") == list(g["vit_keys"])
    m = m.to(dev).eval()
    y = m(torch.from_numpy(g["vit_img"]).to(dev))
    assert y.dtype == torch.bfloat16  # the kernel path ran (fp32 falls back to the torch composition)
    assert rel_l2(y, g["vit_out"]) <= 2e-2
    (y.float() * torch.from_numpy(g["vit_wout"]).to(dev)).sum().backward()
    for k in [k for k in g if k.startswith("grad_")]:
        e = rel_l2(grad_of(m, k[5:]), g[k])
        assert e <= 5e-2, (k, e)


def test_twin_and_decoder_bf16_hip_path_vs_reference_golden(golden, dev, bf16):
    from bridgeqa_amd import med
    g = golden("fusion_med_k.npz")
    t = lambda k: torch.from_numpy(g[k]).to(dev)
    cfg = med.BertConfig(hidden_size=256, num_attention_heads=4, intermediate_size=512, num_hidden_layers=2,
                         vocab_size=200, max_position_embeddings=64, encoder_width=256)
    twin = med.BertModelTwin(config=cfg, add_pooling_layer=False)
    assert keys_of(twin, "text_encoder.") == list(g["twin_keys"])
    twin = twin.to(dev).eval()
    B, P = g["tw_img"].shape[:2]
    img, obj = t("tw_img").requires_grad_(True), t("tw_obj").requires_grad_(True)
    r = twin(t("tw_ids"), attention_mask=t("tw_am"), encoder_hidden_states=img,
             encoder_attention_mask=torch.ones(B, P, dtype=torch.long, device=dev), encoder_hidden_states_twin=obj,
             encoder_attention_mask_twin=t("tw_om"), return_dict=True, output_attentions="last")
    h2d, h3d = r.last_hidden_state
    assert h2d.dtype == torch.bfloat16
    assert rel_l2(h2d, g["tw_h2d"]) <= 2e-2 and rel_l2(h3d, g["tw_h3d"]) <= 2e-2
    assert rel_l2(r.cross_attentions[-1][0], g["tw_cross2d"]) <= 2e-2
    assert rel_l2(r.cross_attentions[-1][1], g["tw_cross3d"]) <= 2e-2
    ((h2d.float() * t("tw_w2")).sum() + (h3d.float() * t("tw_w3")).sum()).backward()
    assert rel_l2(img.grad, g["grad_img"]) <= 5e-2 and rel_l2(obj.grad, g["grad_obj"]) <= 5e-2
    for k in [k for k in g if k.startswith("grad_") and not k.startswith("grad_dec_") and k not in ("grad_img", "grad_obj")]:
        e = rel_l2(grad_of(twin, k[5:]), g[k])
        assert e <= 5e-2, (k, e)
    dec = med.BertLMHeadModel(config=cfg)
    assert keys_of(dec, "text_decoder.") == list(g["dec_keys"])
    dec = dec.to(dev).eval()
    aid = t("dec_ids")
    enc = t("tw_h2d").requires_grad_(True)
    r = dec(aid, attention_mask=t("dec_am"), encoder_hidden_states=enc, encoder_attention_mask=t("tw_am"),
            labels=aid.masked_fill(aid == 0, -100), return_dict=True, reduction="none")
    ref_loss = np.asarray(g["dec_loss"], dtype=np.float64)
    assert np.abs(r.loss.detach().float().cpu().numpy() - ref_loss).max() <= 1e-2 * np.abs(ref_loss).max()
    assert rel_l2(r.logits, g["dec_logits"]) <= 2e-2
    r.loss.sum().backward()
    assert rel_l2(enc.grad, g["grad_dec_enc"]) <= 5e-2
    for k in [k for k in g if k.startswith("grad_dec_") and k != "grad_dec_enc"]:
        e = rel_l2(grad_of(dec, k[9:]), g[k])
        assert e <= 5e-2, (k, e)


def test_blip_vqa3d_bf16_hip_path_vs_reference_golden(golden, dev, bf16):
    """the whole BLIP_VQA3D forward of blip_vqa_3d.py:227-347 (ViT-B width, 2 + 2 layers) through the HIP path: train loss,
    fused states, last-layer attention maps, and the gradients of the step's loss"""
    from bridgeqa_amd.blip_vqa_3d import BLIP_VQA3D, SyntheticTokenizer
    from bridgeqa_amd.med import BertConfig
    g, gg, gm = golden("fusion_blip.npz"), golden("fusion_blip_grad.npz"), golden("fusion_med.npz")
    cfg = BertConfig(num_hidden_layers=2, vocab_size=200, max_position_embeddings=64)
    m = BLIP_VQA3D(med_config=cfg, image_size=64, num_answers=10, use_text_decoder=True, share_decoder=True,
                   scene_size=32, tokenizer=SyntheticTokenizer(0, 102, 198, 199))
    assert keys_of(m, "blip_model.") == list(g["blip_keys"])
    m = m.to(dev).eval()
    t = lambda k: torch.from_numpy(g[k]).to(dev)
    tm = lambda k: torch.from_numpy(gm[k]).to(dev)
    q = {"input_ids": tm("tw_ids"), "attention_mask": tm("tw_am")}
    a = {"input_ids": tm("dec_ids"), "attention_mask": tm("dec_am")}
    img, obj = t("bl_img").requires_grad_(True), t("bl_obj").requires_grad_(True)
    dd = {}
    loss, fused, qmask = m(img, q, a, scene_object_embeds=obj, scene_object_mask=tm("tw_om"), data_dict=dd)
    ref_loss = np.asarray(g["bl_loss"], dtype=np.float64)
    assert np.abs(loss.detach().float().cpu().numpy() - ref_loss).max() <= 1e-2 * np.abs(ref_loss).max()
    assert rel_l2(fused, g["bl_fused"]) <= 2e-2
    assert torch.equal(qmask.cpu(), torch.from_numpy(g["bl_qmask"]))
    assert rel_l2(dd["2d_cross_attention"], g["bl_cross2d"]) <= 2e-2
    assert rel_l2(dd["3d_cross_attention"], g["bl_cross3d"]) <= 2e-2
    (loss.sum() + (fused.float() * torch.from_numpy(gg["bl_wf"]).to(dev)).sum()).backward()
    assert rel_l2(img.grad, gg["grad_img"]) <= 5e-2 and rel_l2(obj.grad, gg["grad_obj"]) <= 5e-2
    for k in [k for k in gg if k.startswith("grad_") and k not in ("grad_img", "grad_obj")]:
        e = rel_l2(grad_of(m, k[5:]), gg[k])
        assert e <= 5e-2, (k, e)
    cand = {"input_ids": t("bl_cand"), "attention_mask": torch.ones_like(t("bl_cand"))}
    with torch.no_grad():
        fused_e, scores, _ = m(t("bl_img"), q, cand, train=False, k_test=3, scene_object_embeds=t("bl_obj"),
                               scene_object_mask=tm("tw_om"), data_dict={})
    assert rel_l2(fused_e, g["bl_fused_eval"]) <= 2e-2
    assert rel_l2(scores[1], g["bl_rank_scene"]) <= 2e-2 and rel_l2(scores[2], g["bl_rank_2d"]) <= 2e-2


def test_twin_levels_stacked_equal_per_stream_path(dev, bf16):
    """the stacked twin level (one grouped launch per projection / LayerNorm for both text streams) against the
    per-stream BertLayer path on the same weights and inputs: states, and every gradient (tolerance = bf16 rounding of
    differently ordered fp32 sums; no dropout)"""
    from bridgeqa_amd import med
    torch.manual_seed(3)
    cfg = med.BertConfig(hidden_size=256, num_attention_heads=4, intermediate_size=512, num_hidden_layers=3,
                         vocab_size=200, max_position_embeddings=64, encoder_width=256)
    twin = med.BertModelTwin(config=cfg, add_pooling_layer=False).to(dev).eval()
    for p in twin.parameters():  # distinct weights in the two streams, non-trivial biases / LayerNorm parameters
        p.data.add_(0.05 * torch.randn_like(p))
    B, L, P2, P3 = 4, 20, 70, 33
    ids = torch.randint(1, 200, (B, L), device=dev)
    am = torch.ones(B, L, dtype=torch.long, device=dev)
    am[1, 15:] = 0
    om = torch.ones(B, P3, dtype=torch.long, device=dev)
    om[2, 20:] = 0
    w2, w3 = torch.randn(B, L, 256, device=dev), torch.randn(B, L, 256, device=dev)
    res = {}
    for mode in (False, True):
        med._TWIN_BATCH[0] = mode
        try:
            img = torch.randn(B, P2, 256, device=dev, generator=torch.Generator(dev).manual_seed(5)).requires_grad_(True)
            obj = torch.randn(B, P3, 256, device=dev, generator=torch.Generator(dev).manual_seed(6)).requires_grad_(True)
            for p in twin.parameters():
                p.grad = None
            r = twin(ids, attention_mask=am, encoder_hidden_states=img,
                     encoder_attention_mask=torch.ones(B, P2, dtype=torch.long, device=dev),
                     encoder_hidden_states_twin=obj, encoder_attention_mask_twin=om, return_dict=True,
                     output_attentions="last")
            h2d, h3d = r.last_hidden_state
            ((h2d.float() * w2).sum() + (h3d.float() * w3).sum()).backward()
            res[mode] = dict(h2d=h2d.float(), h3d=h3d.float(), img=img.grad.float(), obj=obj.grad.float(),
                             grads={n: p.grad.float().clone() for n, p in twin.named_parameters() if p.grad is not None})
        finally:
            med._TWIN_BATCH[0] = True
    a, b = res[False], res[True]
    rel = lambda x, y: ((x - y).norm() / (y.norm() + 1e-20)).item()
    assert rel(b["h2d"], a["h2d"]) <= 5e-3 and rel(b["h3d"], a["h3d"]) <= 5e-3
    assert rel(b["img"], a["img"]) <= 2e-2 and rel(b["obj"], a["obj"]) <= 2e-2
    assert a["grads"].keys() == b["grads"].keys() and len(a["grads"]) > 80
    # a KEY bias shifts every score of a query by the same amount: its gradient is exactly zero in exact arithmetic and
    # what a kernel path computes for it is the sum of its dK rounding errors -- compared on the scale of the VALUE bias'
    # gradient of the same attention (the two paths only agreed on that noise while their dK were bit-identical)
    def err(n):
        if n.endswith(".key.bias"):
            scale = a["grads"][n.replace(".key.bias", ".value.bias")].norm().item()
            return (b["grads"][n] - a["grads"][n]).norm().item() / (scale + 1e-20)
        return rel(b["grads"][n], a["grads"][n])
    worst = sorted(((err(n), n) for n in a["grads"]), reverse=True)[:4]
    assert worst[0][0] <= 2e-2, worst


def test_twin_layer_norm_kernel_equals_two_single_launches(dev):
    from bridgeqa_amd import _ext
    torch.manual_seed(0)
    M, H = 2 * 333, 768
    x = torch.randn(M, H, device=dev).bfloat16()
    r = torch.randn(M, H, device=dev).bfloat16()
    dy = torch.randn(M, H, device=dev).bfloat16()
    ga, ba, gb, bb = [torch.randn(H, device=dev) for _ in range(4)]
    st = torch.full((1,), 11, dtype=torch.int32, device=dev)
    for p in (0.0, 0.1):
        y, mean, rstd, dgb = _ext.twin_drop_add_ln_fwd(x, r, ga, ba, gb, bb, 1e-12, p, 77, st, True)
        assert float(dgb.abs().max()) == 0.0
        dx, dres, dgb = _ext.twin_drop_add_ln_bwd(x, r, ga, gb, dy, mean, rstd, 1e-12, p, 77, st, dgb)
        for g, (gam, bet) in enumerate(((ga, ba), (gb, bb))):
            s = slice(g * M // 2, (g + 1) * M // 2)
            if p == 0.0:  # (the dropout hash is keyed by the absolute row: only p = 0 is comparable launch by launch)
                y1, _, m1, r1, d1 = _ext.drop_add_ln_fwd(x[s].contiguous(), r[s].contiguous(), gam, bet, 1e-12, 0.0, 77, st, False, 0.0, 0, True)
                dx1, dres1, dg1, db1 = _ext.drop_add_ln_bwd(x[s].contiguous(), r[s].contiguous(), gam, dy[s].contiguous(), m1, r1, 1e-12, 0.0, 77, st, None, 0.0, 0, d1)
                assert torch.equal(y[s], y1) and torch.equal(dx[s], dx1) and torch.equal(dres[s], dres1)
                assert torch.allclose(dgb[g, 0], dg1, rtol=1e-4, atol=1e-3) and torch.allclose(dgb[g, 1], db1, rtol=1e-4, atol=1e-3)
            else:
                assert torch.isfinite(y[s].float()).all() and torch.isfinite(dx[s].float()).all()
                kept = (dx[s].float() != 0).float().mean().item()
                assert 0.85 < kept < 0.95, kept


@pytest.mark.parametrize("B,P2,P3,L,D", [(16, 1025, 256, 20, 768), (2, 70, 9, 5, 256), (1, 64, 64, 20, 256)])
def test_twin_kv_equals_concatenated_projections(dev, bf16, B, P2, P3, L, D):
    """ops.twin_kv (reference med.py:549-562 + :112-118: K/V projections of cat(fixed tokens, other stream's states)) reads
    both row sources in place and writes ONE key/value tensor per stream (batched-row maps of bq_gemm_bf16): against the
    torch composition cat -> F.linear in fp32 on the same bf16 operands -- outputs, the fixed tokens' gradient ACCUMULATED
    over two levels in the GradSink, the states' gradient with the tapped share added, weight / bias gradients from two
    row sources, immediate and deferred (grouped flush)."""
    from bridgeqa_amd import fusion_ops as ops
    g = torch.Generator().manual_seed(5)
    mk = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    enc2d, enc3d = mk(B, P2, D).to(torch.bfloat16), mk(B, P3, D).to(torch.bfloat16)
    lins = [[torch.nn.Linear(D, D).to(dev) for _ in range(4)] for _ in range(2)]   # per level: key_a, value_a, key_b, value_b
    for lv in lins:
        for l in lv:
            l.weight.data.copy_(mk(D, D, sc=0.05)); l.bias.data.copy_(mk(D, sc=0.1))
    hs = [mk(2 * B, L, D).to(torch.bfloat16) for _ in range(2)]
    gk = [(mk(B, P2 + L, 2, D).to(torch.bfloat16), mk(B, P3 + L, 2, D).to(torch.bfloat16)) for _ in range(2)]
    gtap = [mk(2 * B, L, D).to(torch.bfloat16) for _ in range(2)]   # what reaches hs through the level's other readers

    def reference():
        e2, e3 = enc2d.float().requires_grad_(True), enc3d.float().requires_grad_(True)
        hl = [h.float().requires_grad_(True) for h in hs]
        ws = [[l.weight.detach().to(torch.bfloat16).float().requires_grad_(True) for l in lv] for lv in lins]
        bs = [[l.bias.detach().clone().requires_grad_(True) for l in lv] for lv in lins]
        outs, loss = [], 0.0
        for lv in range(2):
            m2, m3 = torch.cat((e2, hl[lv][B:]), 1), torch.cat((e3, hl[lv][:B]), 1)
            k2 = torch.stack((F.linear(m2, ws[lv][0], bs[lv][0]), F.linear(m2, ws[lv][1], bs[lv][1])), 2)
            k3 = torch.stack((F.linear(m3, ws[lv][2], bs[lv][2]), F.linear(m3, ws[lv][3], bs[lv][3])), 2)
            outs.append((k2, k3))
            loss = loss + (k2 * gk[lv][0].float()).sum() + (k3 * gk[lv][1].float()).sum() + (hl[lv] * gtap[lv].float()).sum()
        loss.backward()
        return outs, e2.grad, e3.grad, [h.grad for h in hl], [[w.grad for w in wl] for wl in ws], [[b.grad for b in bl] for bl in bs]

    import torch.nn.functional as F
    want = reference()
    rel = lambda a, b: ((a.float() - b.float()).norm() / (b.float().norm() + 1e-12)).item()
    for deferred in (False, True):
        for lv in lins:
            for l in lv:
                l.weight.grad = l.bias.grad = None
        e2, e3 = enc2d.clone().requires_grad_(True), enc3d.clone().requires_grad_(True)
        hl = [h.clone().requires_grad_(True) for h in hs]
        sinks = (ops.GradSink(), ops.GradSink())
        loss, outs = 0.0, []
        for lv in range(2):
            t0 = ops.GradTap()
            k2, k3 = ops.twin_kv(e2, e3, hl[lv], lins[lv][0:2], lins[lv][2:4], sinks[0], sinks[1], tap=t0)
            assert k2.shape == (B, P2 + L, 2, D) and k3.shape == (B, P3 + L, 2, D)
            outs.append((k2, k3))
            loss = loss + (k2.float() * gk[lv][0].float()).sum() + (k3.float() * gk[lv][1].float()).sum() \
                + (ops.tap(hl[lv], t0).float() * gtap[lv].float()).sum()
        if deferred:
            ops.begin_deferred_wgrad()
        loss.backward()
        if deferred:
            ops.flush_deferred_wgrad()
        torch.cuda.synchronize()
        for lv in range(2):
            assert rel(outs[lv][0], want[0][lv][0]) < 6e-3 and rel(outs[lv][1], want[0][lv][1]) < 6e-3
            assert rel(hl[lv].grad, want[3][lv]) < 1e-2, (deferred, lv)
            for j in range(4):
                assert rel(lins[lv][j].weight.grad, want[4][lv][j]) < 1e-2, (deferred, lv, j)
                assert rel(lins[lv][j].bias.grad, want[5][lv][j]) < 1e-2, (deferred, lv, j)
        # (bf16 accumulation over the levels in the sink: one rounding per level)
        assert rel(e2.grad, want[1]) < 1.5e-2 and rel(e3.grad, want[2]) < 1.5e-2, deferred
        assert sinks[0].buf is None and sinks[0].readers == 0


def test_grad_sink_fails_loudly_when_a_level_never_runs_its_backward(dev, bf16):
    """a loss that depends on an intermediate twin level only (output_hidden_states) leaves a reader of the
    GradSink without a backward -- the fixed tokens' gradient used to come back as None with no error"""
    from bridgeqa_amd import fusion_ops as ops
    B, P2, P3, L, D = 2, 70, 9, 5, 256
    g = torch.Generator().manual_seed(3)
    mk = lambda *s: torch.randn(*s, generator=g).to(dev)
    e2, e3 = mk(B, P2, D).to(torch.bfloat16).requires_grad_(True), mk(B, P3, D).to(torch.bfloat16).requires_grad_(True)
    lins = [[torch.nn.Linear(D, D).to(dev) for _ in range(4)] for _ in range(2)]
    hl = [mk(2 * B, L, D).to(torch.bfloat16).requires_grad_(True) for _ in range(2)]
    sinks = (ops.GradSink(), ops.GradSink())
    outs = [ops.twin_kv(e2, e3, hl[lv], lins[lv][0:2], lins[lv][2:4], sinks[0], sinks[1]) for lv in range(2)]
    with pytest.raises(RuntimeError, match="GradSink"):
        (outs[0][0].float().sum() + outs[0][1].float().sum()).backward()   # level 1 never differentiated
    assert sinks[0].buf is None and sinks[0].readers == 0                # (reset: the sink does not poison later passes)


def test_decoder_hoisted_cross_kv_equals_per_layer_projections(dev, bf16):
    """plain encoder / decoder: the key / value projections of all layers' cross-attentions as ONE GEMM over the shared
    encoder states (ops.HoistedKV, gradients written in place into one buffer) against the per-layer projections"""
    from bridgeqa_amd import med
    torch.manual_seed(11)
    cfg = med.BertConfig(hidden_size=256, num_attention_heads=4, intermediate_size=512, num_hidden_layers=3,
                         vocab_size=200, max_position_embeddings=64, encoder_width=256)
    dec = med.BertLMHeadModel(config=cfg).to(dev).eval()
    for p in dec.parameters():
        p.data.add_(0.05 * torch.randn_like(p))
    B, L, Lk = 4, 7, 20
    aid = torch.randint(1, 200, (B, L), device=dev)
    am = torch.ones(B, L, dtype=torch.long, device=dev)
    am[1, 5:] = 0
    em = torch.ones(B, Lk, dtype=torch.long, device=dev)
    em[2, 13:] = 0
    res = {}
    for hoist in (False, True):
        med._HOIST_CROSS_KV = hoist
        try:
            enc = torch.randn(B, Lk, 256, device=dev, generator=torch.Generator(dev).manual_seed(5)).requires_grad_(True)
            for p in dec.parameters():
                p.grad = None
            r = dec(aid, attention_mask=am, encoder_hidden_states=enc, encoder_attention_mask=em,
                    labels=aid.masked_fill(am == 0, -100), return_dict=True, reduction="none", output_attentions=True)
            r.loss.sum().backward()
            res[hoist] = dict(loss=r.loss.detach().float().clone(), logits=r.logits.detach().float().clone(),
                              cross=torch.stack([c.float() for c in r.cross_attentions]),
                              enc=enc.grad.float().clone(),
                              grads={n: p.grad.float().clone() for n, p in dec.named_parameters() if p.grad is not None})
        finally:
            med._HOIST_CROSS_KV = True
    a, b = res[False], res[True]
    rel = lambda x, y: ((x - y).norm() / (y.norm() + 1e-20)).item()
    assert rel(b["logits"], a["logits"]) <= 5e-3 and rel(b["loss"], a["loss"]) <= 5e-3
    assert rel(b["enc"], a["enc"]) <= 2e-2
    assert a["cross"].shape == (3, B, 4, L, Lk) and rel(b["cross"], a["cross"]) <= 5e-3   # maps asked of the hoisted path
    assert a["grads"].keys() == b["grads"].keys()
    assert any("crossattention.self.key.weight" in n for n in a["grads"])
    # a KEY bias shifts every score of a query by the same amount: its gradient is exactly zero in exact arithmetic and
    # what a kernel path computes for it is the sum of its dK rounding errors -- compared on the scale of the VALUE bias'
    # gradient of the same attention (the two paths only agreed on that noise while their dK were bit-identical)
    def err(n):
        if n.endswith(".key.bias"):
            scale = a["grads"][n.replace(".key.bias", ".value.bias")].norm().item()
            return (b["grads"][n] - a["grads"][n]).norm().item() / (scale + 1e-20)
        return rel(b["grads"][n], a["grads"][n])
    worst = sorted(((err(n), n) for n in a["grads"]), reverse=True)[:4]
    assert worst[0][0] <= 2e-2, worst


@pytest.mark.parametrize("optimizer", ["torch", "torch_fused", "bq"])
def test_transposed_weight_copies_follow_the_optimizer(dev, bf16, optimizer):
    """fusion_state.transposed_shadow: the input gradient of a small-M linear / fused QKV / fused MLP read through the
    K-contiguous copy equals the contraction-major read of the operand itself -- bit for bit, on every step of a run in
    which the optimizer keeps changing the weights"""
    ops = bf16
    torch.manual_seed(3)
    lin = torch.nn.Linear(768, 768).to(dev)
    qkv = [torch.nn.Linear(768, 768).to(dev) for _ in range(3)]
    fc1, fc2 = torch.nn.Linear(768, 3072).to(dev), torch.nn.Linear(3072, 768).to(dev)
    params = [p for m in [lin, fc1, fc2] + qkv for p in m.parameters()]
    if optimizer == "bq":
        from bridgeqa_amd.optim import FusedAdamW
        opt = FusedAdamW(params, lr=3e-2)
    else:
        opt = torch.optim.AdamW(params, lr=3e-2, fused=(optimizer == "torch_fused"))

    def dxs(x, g):
        out = []
        for fn in (lambda t: ops.linear(t, lin.weight, lin.bias),
                   lambda t: ops.multi_linear(t, qkv).flatten(-2)[..., :768],
                   lambda t: ops.mlp(t, fc1, fc2)):
            xi = x.clone().requires_grad_(True)
            fn(xi).backward(g)
            out.append(xi.grad)
        return out

    x = torch.randn(16, 20, 768, device=dev).bfloat16()
    g = torch.randn(16, 20, 768, device=dev).bfloat16()
    prev = ops.TRANSPOSED_DX[0]
    try:
        last = None
        for step in range(3):
            opt.zero_grad(set_to_none=True)
            ops.TRANSPOSED_DX[0] = True
            a = dxs(x, g)
            ops.TRANSPOSED_DX[0] = False
            b = dxs(x, g)
            for u, v in zip(a, b):
                assert torch.isfinite(u.float()).all() and torch.equal(u, v), step
            if last is not None:
                assert not torch.equal(a[0], last)   # the weights did move
            last = a[0]
            ops.TRANSPOSED_DX[0] = True
            opt.step()
        assert any(len(k) == 3 for k in ops._TSHADOW)   # the fused QKV operand is registered as one entry
        # an update no optimizer announced (load_state_dict / copy_): the copies follow the version counters
        with torch.no_grad():
            for m in [lin, fc1, fc2] + qkv:
                m.weight.copy_(torch.randn_like(m.weight) * 0.05)
        ops.TRANSPOSED_DX[0] = True
        a = dxs(x, g)
        ops.TRANSPOSED_DX[0] = False
        b = dxs(x, g)
        for u, v in zip(a, b):
            assert torch.equal(u, v)
        assert not torch.equal(a[0], last)
    finally:
        ops.TRANSPOSED_DX[0] = prev

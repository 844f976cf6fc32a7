"""Autograd plumbing of the hoisted K/V projection + two-segment cross-attention (fusion_ops.HoistedKV,
attention_q_kv2) on CPU: the kernels are not involved here (attention_q_kv2 falls back to the concatenated
composition off-GPU), what is checked is that outputs and ALL gradients (fixed tokens, per-layer tails, queries, every
key / value weight and bias) equal the reference wiring  kv_i = [key_i; value_i](cat(x, tail_i))  of med.py:549-562."""
import math

import pytest
import torch


class _SA(torch.nn.Module):
    def __init__(self, hidden):
        super().__init__()
        self.key = torch.nn.Linear(hidden, hidden)
        self.value = torch.nn.Linear(hidden, hidden)


def test_hoisted_kv_two_segment_matches_concatenated_wiring():
    from bridgeqa_amd import fusion_ops as ops
    torch.manual_seed(0)
    hidden, heads, n, B, L1, L2, Lq = 128, 2, 3, 2, 7, 3, 3
    sas = [_SA(hidden) for _ in range(n)]
    x0 = torch.randn(B, L1, hidden)
    tails0 = [torch.randn(B, L2, hidden) for _ in range(n)]
    qs0 = [torch.randn(B, Lq, heads, 64) for _ in range(n)]
    ws = [torch.randn(B, Lq, heads, 64) for _ in range(n)]
    mask = torch.zeros(B, 1, 1, L1 + L2)
    mask[0, 0, 0, 2] = -10000.0
    mask[1, 0, 0, L1 + 1] = -10000.0
    scale = 1.0 / math.sqrt(64)
    prev = ops.set_compute_dtype(torch.bfloat16)
    try:
        def run(hoisted):
            for sa in sas:
                sa.zero_grad()
            x = x0.to(torch.bfloat16).requires_grad_(True)
            tails = [t.to(torch.bfloat16).requires_grad_(True) for t in tails0]
            qs = [q.to(torch.bfloat16).requires_grad_(True) for q in qs0]
            outs = []
            if hoisted:
                hold = ops.HoistedKV(x, sas, heads)
            for i in range(n):
                if hoisted:
                    o = ops.attention_q_kv2(qs[i], hold.kv(i), hold.tail_kv(i, tails[i]), scale, 0.0, mask, sink=(hold, i))
                else:
                    mix = torch.cat((x, tails[i]), dim=1)
                    kv = ops.multi_linear(mix, (sas[i].key, sas[i].value)).view(B, L1 + L2, 2, heads, 64)
                    o = ops.attention_q_kv(qs[i], kv, scale, 0.0, mask)
                outs.append(o)
            loss = sum((o.float() * w).sum() for o, w in zip(outs, ws))
            loss.backward()
            grads = [x.grad.float()] + [t.grad.float() for t in tails] + [q.grad.float() for q in qs]
            for sa in sas:
                grads += [sa.key.weight.grad.float(), sa.value.weight.grad.float(), sa.key.bias.grad.float(),
                          sa.value.bias.grad.float()]
            return [o.detach().float() for o in outs], grads
        o1, g1 = run(True)
        o0, g0 = run(False)
    finally:
        ops.set_compute_dtype(prev)
    rel = lambda a, b: ((a - b).norm() / (b.norm() + 1e-12)).item()
    for a, b in zip(o1, o0):
        assert rel(a, b) < 1e-2
    for k, (a, b) in enumerate(zip(g1, g0)):
        assert a.shape == b.shape and rel(a, b) < 2e-2, (k, rel(a, b))


def test_twin_encoder_hoisted_wiring_matches_concatenated_wiring():
    """BertModelTwin with the hoisted two-segment wiring switched on (med._TWO_SEGMENT) against the default wiring:
    both text streams' outputs and every gradient (image tokens, object tokens, all parameters)."""
    from bridgeqa_amd import fusion_ops as ops, med
    from test_fusion_cpu import small_cfg
    torch.manual_seed(0)
    twin = med.BertModelTwin(config=small_cfg(), add_pooling_layer=False).eval()
    B, L, P, O = 2, 6, 9, 5
    ids = torch.randint(1, 200, (B, L))
    am = torch.ones(B, L, dtype=torch.long); am[1, 4:] = 0
    img0, obj0 = torch.randn(B, P, 64), torch.randn(B, O, 64)
    om = torch.ones(B, O, dtype=torch.long); om[0, 3:] = 0
    prev = ops.set_compute_dtype(torch.bfloat16)
    flag = med._TWO_SEGMENT
    try:
        def run(hoisted):
            med._TWO_SEGMENT = hoisted
            twin.zero_grad()
            img, obj = img0.clone().requires_grad_(True), obj0.clone().requires_grad_(True)
            r = twin(ids, attention_mask=am, encoder_hidden_states=img,
                     encoder_attention_mask=torch.ones(B, P, dtype=torch.long), encoder_hidden_states_twin=obj,
                     encoder_attention_mask_twin=om, return_dict=True, output_attentions="last")  # as BLIP_VQA3D calls it
            h2d, h3d = r.last_hidden_state
            assert len(r.cross_attentions) == 1
            (h2d.float().square().sum() + h3d.float().square().sum()
             + r.cross_attentions[-1][0].float().square().sum()).backward()
            grads = {n: p.grad.float().clone() for n, p in twin.named_parameters() if p.grad is not None}
            return h2d.detach().float(), h3d.detach().float(), img.grad.float(), obj.grad.float(), grads
        a = run(True)
        b = run(False)
    finally:
        med._TWO_SEGMENT = flag
        ops.set_compute_dtype(prev)
    rel = lambda x, y: ((x - y).norm() / (y.norm() + 1e-12)).item()
    for x, y in zip(a[:4], b[:4]):
        assert rel(x, y) < 3e-2, rel(x, y)
    assert a[4].keys() == b[4].keys()
    # key biases have a mathematically zero gradient (softmax is invariant to a per-query shift): rounding noise only
    live = [k for k in a[4] if b[4][k].norm().item() > 1e-3]
    assert len(live) > 40
    worst = max(rel(a[4][k], b[4][k]) for k in live)
    assert max((a[4][k] - b[4][k]).norm().item() for k in a[4] if k not in live) < 1e-3
    assert worst < 6e-2, worst


def test_two_segment_mask_layout():
    """key_mask_log2_two: segment 1's keys at [0, L1), segment 2's at [pad64(L1), pad64(L1) + L2), zeros elsewhere,
    values multiplied by log2(e) (the layout include/bqhip_fusion.h documents for bq_attn_fwd2)"""
    from bridgeqa_amd import _ext
    B, L1, L2 = 2, 70, 3
    mask = torch.zeros(B, 1, 1, L1 + L2)
    mask[0, 0, 0, 5] = -10000.0
    mask[1, 0, 0, L1 + 2] = -10000.0
    m = _ext.key_mask_log2_two(mask, B, L1, L2)
    assert m.shape == (B, 128 + 64) and m.dtype == torch.float32
    want = torch.zeros(B, 192)
    want[0, 5] = -10000.0 * _ext.LOG2E
    want[1, 128 + 2] = -10000.0 * _ext.LOG2E
    assert torch.allclose(m, want)
    one = _ext.key_mask_log2_two(mask[:1], B, L1, L2)  # a (1,1,1,L) mask broadcasts over the batch
    assert torch.allclose(one[1], one[0])

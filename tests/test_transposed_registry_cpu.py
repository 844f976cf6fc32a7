"""fusion_state.transposed_shadow / refresh_transposed: the registry protocol of the K-contiguous weight copies (DESIGN.md
§5.8) on CPU tensors, with the device launch replaced by a host transpose of the same (src, dst) pairs -- registration,
lazy refresh after an optimizer step (the post-step hook), refresh after an unannounced in-place update (version
counters), pruning of dead parameters, refusal of unsupported shapes."""
import gc

import pytest
import torch


@pytest.fixture()
def reg(monkeypatch):
    from bridgeqa_amd import _ext, fusion_ops as ops
    launches = []

    def table(pairs, device):
        return list(pairs), len(pairs)

    def multi(tbl, chunks, max_wgs=0):
        launches.append(len(tbl))
        for src, dst in tbl:
            dst.copy_(src.t())
    monkeypatch.setattr(_ext, "transpose_table", table)
    monkeypatch.setattr(_ext, "transpose_multi", multi)
    saved = (dict(ops._TSHADOW), dict(ops._T_VERSIONS), dict(ops._T_STATE))
    ops._TSHADOW.clear(); ops._T_VERSIONS.clear()
    ops._T_STATE.update(stale=True, tables={}, dirty=True, keep=[], generation=None)
    prev = ops.set_compute_dtype(torch.bfloat16)
    yield ops, launches
    ops.set_compute_dtype(prev)
    ops._TSHADOW.clear(); ops._TSHADOW.update(saved[0])
    ops._T_VERSIONS.clear(); ops._T_VERSIONS.update(saved[1])
    ops._T_STATE.clear(); ops._T_STATE.update(saved[2])


def test_registration_refresh_and_pruning(reg):
    ops, launches = reg
    a, b = torch.nn.Parameter(torch.randn(128, 64)), torch.nn.Parameter(torch.randn(64, 192))
    wa, wb = ops._shadow(a), ops._shadow(b)
    ta = ops.transposed_shadow((a,), wa)
    assert ta.shape == (64, 128) and torch.equal(ta, wa.t()) and not launches       # first use: transposed on the spot
    # marked stale at start: the second operand's first use registers it; the next USE of a registered one refreshes all
    tb = ops.transposed_shadow((b,), wb)
    assert ops.transposed_shadow((a,), wa) is ta and launches == [2]
    assert ops.transposed_shadow((b,), wb) is tb and launches == [2]                 # fresh: no launch
    # an optimizer step: fp32 masters move, the post-step hook re-casts the shadows and marks the copies stale
    opt = torch.optim.SGD([a, b], lr=0.5)
    a.grad, b.grad = torch.ones_like(a), torch.ones_like(b)
    opt.step()
    assert ops._T_STATE["stale"]
    wa2 = ops._shadow(a)
    assert wa2.data_ptr() == wa.data_ptr() and not torch.equal(ta, wa2.t())          # shadow refreshed in place, copy not yet
    assert ops.transposed_shadow((a,), wa2) is ta and torch.equal(ta, wa2.t()) and torch.equal(tb, ops._shadow(b).t())
    assert launches == [2, 2]
    # an update nobody announced (load_state_dict, copy_): seen through the version counters
    with torch.no_grad():
        b.copy_(torch.randn_like(b))
    wb3 = ops._shadow(b)                                                             # (lazy re-cast: a new tensor)
    t3 = ops.transposed_shadow((b,), wb3)
    assert torch.equal(t3, wb3.t())
    # a dead parameter leaves the registry at the next refresh
    del a, wa, wa2, ta, opt
    gc.collect()
    ops.refresh_shadows(only_with_grad=False)                                        # (drops the dead shadow entry)
    ops.mark_transposed_stale()
    ops.transposed_shadow((b,), ops._shadow(b))
    assert len(ops._TSHADOW) == 1 and launches[-1] == 1


def test_unsupported_operands_are_left_to_the_contraction_major_read(reg):
    ops, launches = reg
    p = torch.nn.Parameter(torch.randn(100, 64))          # rows not a multiple of 64
    assert ops.transposed_shadow((p,), ops._shadow(p)) is None
    q = torch.nn.Parameter(torch.randn(128, 64))
    assert ops.transposed_shadow((q,), q.detach()) is None  # fp32 operand
    prev, ops.TRANSPOSED_DX[0] = ops.TRANSPOSED_DX[0], False
    try:
        assert ops.transposed_shadow((q,), ops._shadow(q)) is None
    finally:
        ops.TRANSPOSED_DX[0] = prev
    assert not ops._TSHADOW and not launches


def test_a_generation_owns_its_operands_and_a_reregistration_rewrites_the_copy_in_place(reg):
    """the refresh launch's table holds raw pointers -- the generation a capture used must keep the tensors behind
    them alive, a re-registered operand (load_state_dict: a new bf16 shadow) must not orphan the copy a graph still reads,
    and the history of unowned generations is bounded"""
    ops, launches = reg
    b = torch.nn.Parameter(torch.randn(64, 192))
    wb = ops._shadow(b)
    tb = ops.transposed_shadow((b,), wb)
    ops.refresh_transposed()
    gen = ops.transposed_generation()
    assert gen is not None and any(src.data_ptr() == wb.data_ptr() and dst is tb for src, dst in gen.pairs)
    with torch.no_grad():
        b.copy_(torch.randn_like(b))                     # an update nobody announced: _shadow() re-casts into a NEW tensor
    wb2 = ops._shadow(b)
    assert wb2.data_ptr() != wb.data_ptr()
    tb2 = ops.transposed_shadow((b,), wb2)
    assert tb2 is tb and torch.equal(tb, wb2.t())       # the copy was rewritten in place, not replaced
    ops.refresh_transposed()
    gen2 = ops.transposed_generation()
    assert gen2 is not gen and any(src.data_ptr() == wb2.data_ptr() for src, _ in gen2.pairs)
    assert any(src.data_ptr() == wb.data_ptr() for src, _ in gen.pairs)   # the old generation still owns the old operand
    for _ in range(10):                                  # unowned generations do not pile up
        with torch.no_grad():
            b.copy_(torch.randn_like(b))
        ops.transposed_shadow((b,), ops._shadow(b))
        ops.refresh_transposed()
    assert len(ops._T_STATE["keep"]) <= ops._T_KEEP_MAX

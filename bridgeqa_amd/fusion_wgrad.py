"""Native-GEMM helpers and the DEFERRED, grouped weight gradients of the fusion path (bridgeqa_amd/fusion_ops.py re-exports
everything here): a backward phase computes only the dX chain and parks (dY, X, weight, bias) records; one grouped launch per
tile class produces all dW / db of the phase at its end (reference: the autograd of nn.Linear, models/vit.py:30-32,51-53,
models/med.py:112-118,232,295,310)."""
import os

import torch
import torch.nn.functional as F

from .fusion_state import *  # noqa: F401,F403
from .fusion_state import __all__ as _state_all

# ---- native GEMMs (csrc/gemm.hip through _ext.gemm_*) ------------------------------------------------------------
# Every nn.Linear of the fusion half on the bf16 CUDA path: forward (+bias, +GELU), input gradient (+GELU derivative,
# +bias gradient of the layer before), weight gradient (fp32).  BQ_TORCH_GEMM=1 is a MEASUREMENT knob only: it routes
# the same call sites to torch (hipBLASLt) so that bench.py can A/B the kernels inside the whole step.
_NATIVE_GEMM = [True]   # (tests / tools flip it to run the torch composition of the same node)


def _rows(t):
    """(M, K) bf16 view/copy the GEMM kernels accept: contiguous last dim, 16-B aligned rows"""
    t2 = t.reshape(-1, t.shape[-1])
    if t2.dtype != torch.bfloat16:
        t2 = t2.to(torch.bfloat16)
    if t2.stride(1) != 1 or t2.stride(0) % 8 or t2.data_ptr() % 16:
        t2 = t2.contiguous()
    return t2


def _native_ok(t, N, K):
    """forward y (M, N) = x (M, K) w^T: the contraction K is K-contiguous in both operands => K % 64"""
    return _NATIVE_GEMM[0] and t.is_cuda and compute_dtype() == torch.bfloat16 and K % 64 == 0 and N % 8 == 0


def _native_dx_ok(t, N, K):
    """input gradient dx (M, K) = dy (M, N) w: the contraction is N (K-contiguous in dy) => N % 64"""
    return _NATIVE_GEMM[0] and t.is_cuda and compute_dtype() == torch.bfloat16 and N % 64 == 0 and K % 8 == 0


def _f32_bias(bias):
    """the fp32 master bias itself when possible (no operand copy needed: the kernel adds fp32), else its shadow"""
    if bias is None:
        return None
    if bias.dtype == torch.float32 and bias.is_contiguous():
        return bias.detach()
    return _shadow(bias)


def _mm_f32(a, b):
    """a @ b for bf16 operands with an fp32 result (torch fallback of the weight gradient)"""
    return torch.mm(a.float(), b.float()) if not a.is_cuda else torch.mm(a, b, out_dtype=torch.float32)


def _dw_db(g2, x2, need_dw, need_db):
    """weight / bias gradient of one linear, now: dW = g2^T x2 (fp32), db = column sums of g2 (fp32)"""
    dw = db = None
    native = g2.is_cuda and g2.dtype == torch.bfloat16 and x2.dtype == torch.bfloat16 and g2.shape[1] % 8 == 0 \
        and x2.shape[1] % 8 == 0 and _NATIVE_GEMM[0]
    if need_dw:
        if native:
            from . import _ext
            dw = _ext.gemm_dw(_rows(g2), _rows(x2), tile=256 if g2.shape[0] >= _BIG_ROWS else 64)
        else:
            dw = _mm_f32(g2.t(), x2)
    if need_db:
        if native:
            from . import _ext
            db = _ext.colsum_grouped([_rows(g2)])[0]
        else:
            db = g2.sum(0, dtype=torch.float32)
    return dw, db


# ---- deferred, grouped weight gradients ---------------------------------------------------------------------------
# dW / db are not on the critical path of a backward pass.  Inside a begin/flush scope the backward of every bf16
# linear computes ONLY dX and parks (dY, X, parameters); flush_deferred_wgrad() then produces ALL weight gradients
# of the scope with ONE grouped GEMM launch per tile class (12 ViT blocks x 4 linears = 48 problems, 1296 tiles of
# 256 x 256 with the full 16400-row contraction each -- instead of 48 launches of 27-36 tiles) and ALL bias gradients
# with one grouped column-sum launch.  MI355X has the HBM for it: the parked dY of config c3 are 2.7 GB.
_DEFER = [None]
_BIG_ROWS = 1024  # contractions at least this long run on the 256 x 256 kernel (measured: sending the text side's 160-640-row
                  # contractions there too costs +1.5 ms per c3 step -- its pipeline fill and 256 KB fp32 tile epilogue dominate)


def begin_deferred_wgrad():
    _DEFER[0] = []


def _accumulate_grad(p, g):
    if p is None or not p.requires_grad:
        return
    if p.grad is None:
        p.grad = g
    else:
        p.grad.add_(g)


def take_deferred_wgrad():
    """end the scope WITHOUT computing: returns the parked (dY, X, weights, biases) records for
    flush_deferred_items() -- pipeline.PhasedTrainStep produces the fusion phase's weight gradients on another stream,
    off the critical path between the fusion backward and the image / detector backward"""
    items, _DEFER[0] = _DEFER[0], None
    return items or []


def flush_deferred_wgrad():
    """compute the parked weight / bias gradients (current stream) and store them in the parameters' .grad"""
    flush_deferred_items(take_deferred_wgrad())


_PLAN_CACHE = {}
_PLAN_WGRAD = [True]
_QSUM = [True]
_QSUM64 = [True]
_DW_TILE = [256]   # 256: gemm256_kernel (one workgroup per CU); 128: the 256 x 128 persistent kernel's weight-gradient form
_SHORT_DW_MIN_ROWS = [64]    # contractions from this many rows on (the decoder's B x 5 answer rows = 80 included: the flush 0.53 -> 0.48 ms,
                             # bit-identical; the 256 x 128 form, _SHORT_DW_TILE = 128, needs >= 128)
_SHORT_DW_TILE = [256]  # short contractions (128 <= rows < 1024: the text side's B x 20 token rows): 256 = gemm256_kernel (since
                        # round 6's K loop: the 228 problems of one flush 0.52 ms against 0.59 on the persistent 256 x 128
                        # kernel's weight-gradient form and 0.77 on the 64 x 64-tile kernel, tools/bench_short_dw.py, each form
                        # graph-replayed and interleaved, bit-identical results; c3 step 31.31 -> 31.17 ms over five interleaved
                        # pairs, profiles/r06_short_dw_tile.txt); 128 and 64 select the other two


def plan_big_launches(tiles, cus, max_problems=36, moved_cost=0.011):
    """How to issue the 256 x 256-tile weight-gradient problems of one flush.  Every tile of such a launch runs the full
    contraction (all rows of the batch), so a launch costs ceil(tiles / cus) rounds of equal length -- 1296 tiles on 256 CUs
    are 5.06 rounds, i.e. SIX, for sixteen tiles.  Returns (groups, moved): `groups` = lists of problem indices, one grouped
    launch each (<= max_problems problems, the kernel-argument limit), `moved` = indices of small problems sent to the
    64 x 64-tile kernel instead, chosen so that rounds + moved_cost * (tiles moved) is smallest.  The problems come in a
    handful of distinct sizes, so the search walks the counts per size of the second launch (a few thousand cases, cached)."""
    key = (tuple(tiles), cus, max_problems)
    hit = _PLAN_CACHE.get(key)
    if hit is not None:
        return hit
    n = len(tiles)
    rounds = lambda t: -(-t // cus) if t > 0 else 0
    order = sorted(range(n), key=lambda k: tiles[k])
    default = [list(range(n))[i:i + max_problems] for i in range(0, n, max_problems)]
    best = (sum(rounds(sum(tiles[k] for k in g)) for g in default), default, [])
    if n <= 2 * max_problems:
        for m in range(0, min(6, n)):
            moved = order[:m]
            rest = order[m:]
            cost_moved = moved_cost * sum(tiles[k] for k in moved)
            by_size = {}
            for k in rest:
                by_size.setdefault(tiles[k], []).append(k)
            sizes = sorted(by_size)
            if len(sizes) > 5:
                break
            total, cnt = sum(tiles[k] for k in rest), len(rest)
            combos = [[]]
            for sz in sizes:
                combos = [c + [x] for c in combos for x in range(len(by_size[sz]) + 1)]
                if len(combos) > 50000:
                    combos = None
                    break
            if combos is None:
                break
            for c in combos:
                nb = sum(c)
                if nb > max_problems or cnt - nb > max_problems:
                    continue
                tb = sum(x * sz for x, sz in zip(c, sizes))
                cost = rounds(tb) + rounds(total - tb) + cost_moved
                if cost < best[0] - 1e-9:
                    gb = [k for x, sz in zip(c, sizes) for k in by_size[sz][:x]]
                    ga = [k for x, sz in zip(c, sizes) for k in by_size[sz][x:]]
                    best = (cost, [g for g in (ga, gb) if g], list(moved))
    plan = (best[1], best[2])
    _PLAN_CACHE[key] = plan
    return plan


def _nrows(t):
    """rows of a GEMM row operand: (rows, cols), or a (batch, rows, cols) batched-row view (_ext._mat_rows)"""
    return t.shape[0] if t.dim() == 2 else t.shape[0] * t.shape[1]


def flush_deferred_items(items):
    """items: (dY, X, weights, biases[, primary]) records.  `primary` (optional) = index of ANOTHER record of the same
    weights: this record's rows are a second row source of that weight gradient (the twin K/V projection reads image
    tokens and the other stream's states, _TwinKVFn) and are ADDED to it by a launch behind the primary one."""
    if not items:
        return
    from . import _ext
    flags = _ext.GEMM_P_XC | _ext.GEMM_Q_XC | _ext.GEMM_OUT_F32
    items = [tuple(it) + (None,) * (5 - len(it)) for it in items]
    dws = [None] * len(items)
    prim = [k for k, it in enumerate(items) if it[4] is None]
    second = [k for k, it in enumerate(items) if it[4] is not None]
    big = [k for k in prim if _nrows(items[k][0]) >= _BIG_ROWS]
    small = [k for k in prim if _nrows(items[k][0]) < _BIG_ROWS]
    if _PLAN_WGRAD[0]:
        # longest contractions first: a launch's workgroups start in tile order, so the long tiles (16 720-row image
        # tokens) run from the beginning and the short ones (4 416-row object tokens) fill in behind them (A/B x4: 40.88 vs
        # 40.91 ms -- inside the noise; kept because it cannot hurt)
        big.sort(key=lambda k: -_nrows(items[k][0]))
    groups = [big] if big else []
    if big and _PLAN_WGRAD[0]:
        tj = _DW_TILE[0]
        tiles = [-(-items[k][0].shape[-1] // tj) * -(-items[k][1].shape[-1] // 256) for k in big]
        cus = torch.cuda.get_device_properties(items[big[0]][0].device).multi_processor_count
        plan_groups, moved = plan_big_launches(tiles, cus * (256 // tj))
        groups = [[big[j] for j in g] for g in plan_groups]
        small = small + [big[j] for j in moved]
    dbs = {}
    mid = []
    if _SHORT_DW_TILE[0] in (128, 256):
        min_rows = max(_SHORT_DW_MIN_ROWS[0], 128 if _SHORT_DW_TILE[0] == 128 else 1)
        # (only the SHORT contractions: the planner's moved problems -- full 16 400-row contractions of 9 tiles -- stay on
        # the 64-tile kernel, whose cut contraction spreads them over the chip: the 256 x 128 form ran them at 5 % MFMA busy)
        ok = lambda it: (min_rows <= _nrows(it[0]) < _BIG_ROWS and it[0].dim() == 2 and it[0].shape[-1] >= 128
                         and it[1].shape[-1] >= 256)
        mid = [k for k in small if ok(items[k])]
        small = [k for k in small if not ok(items[k])]
    for tile, idx_groups in ((_DW_TILE[0], groups), (_SHORT_DW_TILE[0], [mid] if mid else []), (64, [small] if small else [])):
        for idx in idx_groups:
            probs = []
            for k in idx:
                g2, x2 = items[k][0], items[k][1]
                dws[k] = torch.empty(g2.shape[-1], x2.shape[-1], dtype=torch.float32, device=g2.device)
                pr = dict(P=x2, Q=g2, out=dws[k])
                if items[k][3] is not None and _QSUM[0] and (tile != 64 or _QSUM64[0]):
                    # the bias gradient (column sums of dY) from the same launch: four more MFMAs per K tile in a third
                    # of the workgroups instead of a second pass over dY (csrc/gemm.hip, QSUM)
                    dbs[k] = torch.empty(g2.shape[-1], dtype=torch.float32, device=g2.device)
                    pr["colsum"] = dbs[k]
                probs.append(pr)
            _ext.gemm_grouped(probs, flags, _ext.EPI_NONE, tile)
    with_b = [k for k in prim if items[k][3] is not None and k not in dbs]
    if with_b:
        dbs.update(zip(with_b, _ext.colsum_grouped([items[k][0].reshape(-1, items[k][0].shape[-1]) for k in with_b])))
    if second:
        # second row sources: added onto the stored gradient of their primary record (same stream: ordered behind it) by
        # the small-tile kernel's atomic epilogue -- short contractions (B x 20 text rows)
        probs = []
        for k in second:
            g2, x2, pk = items[k][0], items[k][1], items[k][4]
            pr = dict(P=x2, Q=g2, out=dws[pk], accum=True)
            if items[pk][3] is not None:
                pr["colsum"] = dbs[pk]
            probs.append(pr)
        _ext.gemm_grouped(probs, flags, _ext.EPI_NONE, 64)
    for k in prim:
        g2, x2, ws, bs = items[k][:4]
        n = g2.shape[-1] // len(ws)
        for j, w in enumerate(ws):
            _accumulate_grad(w, dws[k][j * n:(j + 1) * n] if len(ws) > 1 else dws[k].view(w.shape))
            if bs is not None:
                _accumulate_grad(bs[j], dbs[k][j * n:(j + 1) * n] if len(ws) > 1 else dbs[k])


def _defer_ok(g2, x2):
    return (_DEFER[0] is not None and _NATIVE_GEMM[0] and g2.is_cuda and g2.dtype == torch.bfloat16
            and x2.dtype == torch.bfloat16 and g2.shape[1] % 8 == 0 and x2.shape[1] % 8 == 0)


def _park(g2, x2, ws, bs):
    _DEFER[0].append((_rows(g2), _rows(x2), ws, bs))


def _park_two(g_a, x_a, g_b, x_b, ws, bs):
    """one weight gradient from TWO row sources: dW = g_a^T x_a + g_b^T x_b (g_*: (rows, N) or batched-row views
    (batch, rows, N) read in place; x_*: (rows, K)); the second source is added behind the first (flush_deferred_items)"""
    d = _DEFER[0]
    d.append((g_a, _rows(x_a), ws, bs))
    d.append((g_b, _rows(x_b), ws, bs, len(d) - 1))



__all__ = [n for n in list(globals()) if not n.startswith("__")]

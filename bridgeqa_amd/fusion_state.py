"""State shared by the dense operators of the fusion path (bridgeqa_amd/fusion_ops.py is the public surface and re-exports
everything here): the compute dtype, the side streams / fork helper of the stream-level concurrency, and the bf16
operand copies ("shadows") of the fp32 parameters with their refresh protocol (optimizer hook, FusedAdamW writes them
in its own pass).  No autograd node lives here."""
import math

import os

import torch
import torch.nn.functional as F

_COMPUTE_DTYPE = torch.float32

# ---- stream-level concurrency --------------------------------------------------------------------
# Large parts of the step are chains of short, latency-bound kernels that use a fraction of the 256 CUs
# (furthest point sampling runs on B workgroups; the text streams work on 320 tokens).  Independent chains are
# therefore issued on separate HIP streams -- detector branch || image encoder, 2D text stream || 3D text stream --
# and joined with events; under HIP-graph capture the forks become parallel branches of the graph.
_OVERLAP = [True]
SHAREDMLP_BF16 = [False]
# Detector fast path (bf16 compute dtype only): features are kept POINT-MAJOR (B,N,C) so neighbourhood grouping is
# a copy of contiguous rows, the grouped tensor is written once as bf16 NHWC and the SharedMLP convolutions run on
# it without layout transposes.  Values at the module boundary keep the reference layout (B,C,N) as strided views.
POINT_MAJOR = [True]
_SIDE_STREAMS = {}


def set_overlap(flag):
    prev, _OVERLAP[0] = _OVERLAP[0], bool(flag)
    return prev


def overlap_enabled(t):
    return _OVERLAP[0] and t.is_cuda


def side_stream(name, device):
    key = (name, device.index if device.index is not None else torch.cuda.current_device())
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]


_PHASE_STREAMS = {}


def phase_streams(device, main_priority=-1, det_priority=0):
    """(main, detector) phase streams of `device`, created ONCE per process and priority pair: pipeline.PhasedTrainStep and
    graphed.GraphedRunner replay their phase graphs on these.  Streams are a scarce per-process resource on this runtime (a
    handful of hardware queues; streams beyond them share queues, and two phase streams that land on one queue execute
    their graphs packet by packet behind each other): a second runner in the same process -- bench.py's reference-loop
    measurement after the phased one -- ran its latency-bound phases 3x slower on freshly created streams (fusion forward
    11.1 against 3.9 ms) than on the first runner's."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), int(main_priority), int(det_priority))
    if key not in _PHASE_STREAMS:
        _PHASE_STREAMS[key] = (torch.cuda.Stream(device=device, priority=int(main_priority)),
                               torch.cuda.Stream(device=device, priority=int(det_priority)))
    return _PHASE_STREAMS[key]


class fork(object):
    """with fork("name", tensor) as s: ... runs the body on a side stream that first waits for the current one.
    Call .join(*tensors) afterwards: the current stream waits for the side stream and the tensors produced on it
    are marked as used by the current stream (allocator safety)."""

    def __init__(self, name, like):
        self.main = torch.cuda.current_stream(like.device)
        self.side = side_stream(name, like.device)
        self.ctx = torch.cuda.stream(self.side)

    def __enter__(self):
        self.side.wait_stream(self.main)
        self.ctx.__enter__()
        return self

    def __exit__(self, *a):
        return self.ctx.__exit__(*a)

    def uses(self, *tensors):
        """tensors made on the main stream that the side stream reads"""
        for t in tensors:
            if t is not None:
                t.record_stream(self.side)

    def join(self, *tensors):
        self.main.wait_stream(self.side)
        for t in tensors:
            if t is not None:
                t.record_stream(self.main)


_OPT_HOOK = [None]


def set_compute_dtype(dtype):
    """bf16: GEMM operands are bf16 copies ("shadows") of the fp32 master parameters.  A global optimizer post-step
    hook keeps them current (refresh_shadows), whatever optimizer the caller uses."""
    global _COMPUTE_DTYPE
    prev, _COMPUTE_DTYPE = _COMPUTE_DTYPE, dtype
    if dtype != torch.float32 and _OPT_HOOK[0] is None:
        from torch.optim.optimizer import register_optimizer_step_post_hook
        # (optim.FusedAdamW writes most shadows in its own kernel and lists them in `shadow_ids`: they are never re-cast
        # here -- without this a SUBSET step re-cast the other subsets' 399 M weights, 2.3 ms)
        _OPT_HOOK[0] = register_optimizer_step_post_hook(
            lambda opt, args, kwargs: (refresh_shadows(skip=getattr(opt, "shadow_ids", None), only=_opt_param_ids(opt)),
                                       mark_transposed_stale()))
    return prev


def _opt_param_ids(opt):
    """ids of the parameters `opt` steps (cached on it): the post-step hook refreshes THEIR shadows only -- a second model
    alive in the process (tests; an EMA copy) must not have its shadows re-cast, least of all inside another model's
    graph capture"""
    ids = getattr(opt, "_bq_param_ids", None)
    n = sum(len(g["params"]) for g in opt.param_groups)
    if ids is None or ids[0] != n:
        ids = (n, frozenset(id(p) for g in opt.param_groups for p in g["params"]))
        opt._bq_param_ids = ids
    return ids[1]


def compute_dtype():
    return _COMPUTE_DTYPE


def _c(t):
    """Cast to the compute dtype (inside autograd: gradients reach an fp32 parameter through the cast).  No caching:
    version counters cannot be trusted to see an optimizer update (the fused multi-tensor optimizers do not bump
    them -- measured on torch 2.10), and the parameters on hot paths go through _shadow() / refresh_shadows()."""
    if t.dtype == _COMPUTE_DTYPE:
        return t
    return t.to(_COMPUTE_DTYPE)


_SHADOW = {}


def _shadow(t):
    """bf16 copy of an fp32 parameter made OUTSIDE autograd, once per optimizer step (keyed by the in-place
    version counter).  _LinearFn routes the gradient to the fp32 parameter itself, in fp32."""
    if t.dtype == _COMPUTE_DTYPE:
        return t.detach()
    hit = _SHADOW.get(id(t))
    if hit is not None and hit[0]() is t and hit[1] == t._version and hit[2].dtype == _COMPUTE_DTYPE:
        return hit[2]
    import weakref
    with torch.no_grad():
        c = t.detach().to(_COMPUTE_DTYPE)
    _SHADOW[id(t)] = (weakref.ref(t), t._version, c)
    return c


def shadow_of(param):
    """the registered bf16 shadow tensor of a parameter that an optimizer kernel may write element for element (None if
    it has none, is not of the current compute dtype, or is a strided view -- those are refreshed by refresh_shadows)"""
    ent = _SHADOW.get(id(param))
    if (ent is None or ent[0]() is not param or ent[2].dtype != _COMPUTE_DTYPE or ent[2].device != param.device
            or not ent[2].is_contiguous()):
        return None
    return ent[2]


_PADDED = {}


def padded_conv_shadow(weight):
    """bf16 GEMM operand of a 1x1 convolution weight (N, K, 1, 1): (N, Kc) with Kc = K rounded up to 64, ZERO beyond K
    (csrc/gemm.hip pwconv64_kernel contracts whole 64-wide K tiles; the padding multiplies whatever follows the K
    channels of a point row).  The registered shadow of the parameter is the (N, K, 1, 1) view of that buffer, so the
    optimizer's shadow refresh keeps it current."""
    import weakref
    N, K = weight.shape[0], weight.shape[1]
    hit = _PADDED.get(id(weight))
    if hit is not None and hit[0]() is weight and hit[1].device == weight.device:
        ent = _SHADOW.get(id(weight))
        if ent is not None and ent[2].untyped_storage().data_ptr() == hit[1].untyped_storage().data_ptr():
            if ent[1] != weight._version:  # an in-place update nobody refreshed (load_state_dict, plain optimizers)
                with torch.no_grad():
                    ent[2].copy_(weight.detach())
                _SHADOW[id(weight)] = (ent[0], weight._version, ent[2])
            return hit[1]
    Kc = (K + 63) // 64 * 64
    with torch.no_grad():
        buf = torch.zeros(N, Kc, dtype=torch.bfloat16, device=weight.device)
        view = buf[:, :K]
        for _ in range(weight.dim() - 2):  # (N, K, 1, 1) for Conv2d, (N, K, 1) for Conv1d: a view of the padded rows
            view = view.unsqueeze(-1)
        view.copy_(weight.detach())
    _SHADOW[id(weight)] = (weakref.ref(weight), weight._version, view)
    _PADDED[id(weight)] = (weakref.ref(weight), buf)
    return buf


_FRESH = set()


def shadows_written(params):
    """An optimizer that writes the shadows itself (optim.FusedAdamW) reports them here; the post-step hook's
    refresh_shadows() then skips them."""
    for p in params:
        ent = _SHADOW.get(id(p))
        if ent is not None:
            _SHADOW[id(p)] = (ent[0], p._version, ent[2])
            _FRESH.add(id(p))


def refresh_shadows(only_with_grad=True, skip=None, only=None):
    """Bring every registered bf16 shadow (and, through them, the concatenated QKV / KV operands, whose row blocks ARE
    the per-weight shadows) up to date with ONE multi-tensor cast.  Runs as a global optimizer post-step hook
    (set_compute_dtype); call it yourself after any other in-place parameter update.  It does NOT consult version
    counters: torch's fused multi-tensor optimizers update parameters without bumping them (the lazy check in
    _shadow() only catches ordinary in-place ops and load_state_dict).  only_with_grad: skip parameters that have no
    gradient, i.e. that the optimizer did not touch.  skip: ids of parameters whose shadows the stepping optimizer writes
    itself.  only: ids of the parameters to consider (the stepping optimizer's own)."""
    dst, src = [], []
    for key, (ref, ver, c) in list(_SHADOW.items()):
        t = ref()
        if t is None:
            del _SHADOW[key]
            continue
        if only is not None and key not in only:
            continue
        if skip is not None and key in skip and ver == t._version:
            continue
        if c.dtype != _COMPUTE_DTYPE or c.device != t.device:
            continue
        if key in _FRESH and ver == t._version:
            continue  # written by the optimizer kernel itself
        if only_with_grad and t.grad is None and ver == t._version:
            continue
        dst.append(c)
        src.append(t.detach())
        _SHADOW[key] = (ref, t._version, c)
    if only is None:
        _FRESH.clear()
    else:
        _FRESH.difference_update(only)
    if dst:
        with torch.no_grad():
            torch._foreach_copy_(dst, src)
    return len(dst)



# ---- transposed shadows: the K-contiguous operand of the small-M input-gradient GEMMs ---------------------------------
# dX = dY W contracts over the ROWS of the (N, K) weight operand.  Read contraction-major, the text side's small launches
# (80-640 rows: latency chains) take up to twice as long as their forward twins (csrc/transpose.hip header); a second
# bf16 copy W^T (K, N) lets them run on the forward's operand form.  Entries are keyed by the parameters a fused operand
# is made of; all of them are refreshed by ONE launch -- lazily at the first use after an optimizer step, or explicitly
# (refresh_transposed) by a caller that wants the launch somewhere else (pipeline.PhasedTrainStep: the text_prep phase,
# beside the image encoder; under graph replay that captured launch is the refresh).
TRANSPOSED_DX = [True]
_TSHADOW = {}
_T_VERSIONS = {}
_T_STATE = {"stale": True, "tables": {}, "dirty": True, "keep": []}
_T_KEEP_MAX = 4   # generations kept alive without an owner (eager launches in flight, hand-made captures)


class TransposedGeneration(object):
    """One build of the refresh launch's device tables TOGETHER with the (src, dst) tensors its records point at.  A capture
    that recorded the launch holds the generation it used (transposed_generation(): pipeline.PhasedTrainStep and
    graphed.GraphedRunner keep it next to their graphs), so the operands a replayed refresh reads and writes cannot return to
    the caching allocator while the graph lives -- a re-registration (load_state_dict allocates a new bf16 shadow) only makes
    a NEW generation.  `_T_STATE["keep"]` holds the last few generations for launches nobody owns."""

    def __init__(self, tables, pairs):
        self.tables, self.pairs = tables, pairs


def transposed_generation():
    """the current generation object (None before the first build): keep a reference for as long as a captured graph may
    replay the refresh launch"""
    return _T_STATE.get("generation")


def _cast_versions(params):
    """the parameter versions the bf16 shadows (what the copies are transposed FROM) were cast at"""
    out = []
    for p in params:
        ent = _SHADOW.get(id(p))
        out.append(ent[1] if (ent is not None and ent[0]() is p) else p._version)
    return tuple(out)


def mark_transposed_stale():
    _T_STATE["stale"] = True


def transposed_shadow(params, wb):
    """(K, N) bf16 contiguous transpose of the (N, K) operand `wb` built from the fp32 parameters `params` (a tuple: the
    k fused linears whose shadows are wb's row blocks), kept current across optimizer steps; None when unavailable (odd
    shapes, or a first use inside a stream capture -- the caller then reads wb contraction-major)."""
    if not TRANSPOSED_DX[0] or wb.dim() != 2 or wb.shape[0] % 64 or wb.shape[1] % 64 or wb.stride(1) != 1 \
            or wb.stride(0) % 8 or wb.dtype != torch.bfloat16:
        return None
    key = tuple(id(p) for p in params)
    ent = _TSHADOW.get(key)
    capturing = wb.is_cuda and torch.cuda.is_current_stream_capturing()
    if ent is not None and all(r() is p for r, p in zip(ent[0], params)) and ent[1].data_ptr() == wb.data_ptr() \
            and ent[1].shape == wb.shape:
        # an in-place update nobody announced (load_state_dict, a plain copy_): the version counters moved since the last
        # refresh -- fused optimizers do not bump them, which is what the post-step hook's mark_transposed_stale() is for
        if not capturing and _T_VERSIONS.get(key) != tuple(p._version for p in params):
            _T_STATE["stale"] = True
        if _T_STATE["stale"]:
            # (inside a stream capture the refresh launch becomes part of the graph -- as it must: the replayed step needs
            # it too; a registry that changed since its device table was built cannot be rebuilt there)
            if capturing and _T_STATE["dirty"]:
                return None
            refresh_transposed()
        return ent[2]
    if capturing:
        return None
    import weakref
    with torch.no_grad():
        if ent is not None and all(r() is p for r, p in zip(ent[0], params)) and ent[2].shape == (wb.shape[1], wb.shape[0]) \
                and ent[2].device == wb.device:
            # the same parameters behind a NEW operand tensor (load_state_dict: _shadow() re-cast into a fresh tensor): the
            # existing copy is rewritten in place -- whoever holds it (a captured graph's dX launches) keeps reading a live,
            # current tensor; only the refresh table's source pointer changes (a new generation)
            wt = ent[2]
            wt.copy_(wb.t())
        else:
            wt = wb.t().contiguous()
    _TSHADOW[key] = ([weakref.ref(p) for p in params], wb.detach(), wt)
    _T_VERSIONS[key] = _cast_versions(params)
    _T_STATE["dirty"] = True
    return wt


def refresh_transposed(device=None, max_wgs=0):
    """re-transpose every registered operand (one launch per device); called with the shadows already current on the
    calling stream (after the optimizer step / refresh_shadows).  max_wgs: see _ext.transpose_multi"""
    from . import _ext
    capturing = torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
    dead = [] if capturing else [k for k, e in _TSHADOW.items() if any(r() is None for r in e[0])]
    if capturing and _T_STATE["dirty"]:
        raise RuntimeError("refresh_transposed: operands were registered since the last refresh; run one eager step (or "
                           "call refresh_transposed()) before capturing")
    if _T_STATE["dirty"] or dead:
        for key in dead:
            del _TSHADOW[key]
            _T_VERSIONS.pop(key, None)
        by_dev = {}
        for ent in _TSHADOW.values():
            by_dev.setdefault(ent[1].device, []).append((ent[1], ent[2]))
        # the previous generation (tables AND the tensors they point at) stays alive for whoever captured its launch
        # (TransposedGeneration); without an owner it lives on in a bounded history
        if _T_STATE.get("generation") is not None:
            _T_STATE["keep"].append(_T_STATE["generation"])
            del _T_STATE["keep"][:-_T_KEEP_MAX]
        _T_STATE["tables"] = {dev: _ext.transpose_table(pairs, dev) for dev, pairs in by_dev.items()}
        _T_STATE["generation"] = TransposedGeneration(_T_STATE["tables"], [pr for prs in by_dev.values() for pr in prs])
        _T_STATE["dirty"] = False
    skipped = False
    for dev, (table, chunks) in _T_STATE["tables"].items():
        if device is None or dev == device or (device.index is None and dev.type == device.type):
            _ext.transpose_multi(table, chunks, max_wgs)   # (sets the device of `table` itself)
        else:
            skipped = True   # (several devices in one process: the others' copies stay stale)
    if not capturing:
        for key, ent in _TSHADOW.items():
            ps = [r() for r in ent[0]]
            if all(p is not None for p in ps):
                _T_VERSIONS[key] = _cast_versions(ps)
    _T_STATE["stale"] = skipped


__all__ = [n for n in list(globals()) if not n.startswith("__")]

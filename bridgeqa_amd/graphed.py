"""HIP-graph replay BEHIND the reference's plain training loop.

The reference trains with (lib/solver.py:463-595 `_forward` / `_backward`, scripts/train.py:410-417):

    data_dict = model(data_dict)            # ScanQA.forward
    _, data_dict = get_loss(data_dict, ...) # lib/loss_helper.py, outside the model
    optimizer.zero_grad(); data_dict["loss"].backward(); clip_grad_value_(...); optimizer.step()

Run kernel by kernel, that sequence is host-launch bound on this path (61-66 ms per c3 step against 37-38 ms for
pipeline.PhasedTrainStep, which wants the solver rewritten around it).  `enable(model)` keeps the loop as it is and puts
graphs behind the two calls that carry the work:

  * `model(data_dict)` in train mode copies the inputs into static buffers and REPLAYS three captured graphs -- detector
    forward on a second stream beside the image encoder's forward, then the twin fusion + answer decoder forward -- and
    returns a data_dict whose tensors alias the graphs' static outputs, tied to ONE autograd node;
  * `loss.backward()` reaches that node with the gradients of whatever outputs the caller's loss used; the node copies
    them into static gradient buffers and REPLAYS the backward graphs -- fusion backward (+ its grouped weight-gradient
    flush), then detector backward beside the image encoder's backward -- which leave every parameter's gradient in
    static memory; `p.grad` is pointed at it (an `optimizer.zero_grad()` before the backward may have set it to None).
  * the loss itself and `optimizer.step()` stay the caller's, launched eagerly: the host enqueues them while the GPU is
    still busy with the replayed forward, so their launch latency hides.

What the captured path assumes (checked where it can be): fixed input shapes (a change re-captures), one backward per
forward, gradients OVERWRITTEN per step (no accumulation over several backward calls; the reference zeroes them every
iteration), outputs valid until the next forward (they are rewritten in place); at the FIRST graphed forward no autograd
graph of an earlier eager forward may still be alive (drop the previous outputs / loss: its AccumulateGrad nodes were born on
the caller's stream and this runtime's hipStreamEndCapture faults on them).  Eval mode and `torch.no_grad()` take the
ordinary eager path.

Round 5 additions, all outside the loop BODY:
  * `prefetch_loader(model, dataloader)` around the loop's iterable (or `model._graphed.prefetch(next_point_clouds)`): the
    NEXT batch's sampling / grouping indices (FPS, ball query, three-NN: coordinates only, no parameters) run on the
    detector stream under THIS step's fusion, as pipeline.PhasedTrainStep does; without an announcement a step computes its
    own geometry in front of its detector forward, as before.
  * `wrap_optimizer(model, optimizer)`: `optimizer.step()` replays a captured graph of the fused update (and
    `optimizer.zero_grad()` launches nothing: the captured backward overwrites every gradient).
  * data parallel (`torchrun`, scripts/train.py:346-347 wraps the model in DistributedDataParallel): the replayed backward
    runs no AccumulateGrad node, so DDP's reducer hooks would never fire and the replicas would silently diverge.
    `enable()` therefore refuses a DDP-wrapped model and, when a process group with more than one rank is initialised,
    exchanges the static gradients itself (ddp.PackedGradReducer: the fusion group under the image / detector backward,
    the rest at the end of the backward), broadcasts rank 0's parameters once and its buffers before every forward.
"""
import torch

from . import fusion_ops as ops


_INPUT_KEYS = ("point_clouds", "images", "question", "answer")
# data_dict entries (host values) that change every iteration and that the loss does not depend on: left out of wrap_loss'
# capture signature.  Callers whose loop adds more of them extend the set.
VOLATILE_KEYS = {"iteration"}


def enable(model, warmup=2, optimizer=None, comm_dtype=torch.float32, process_group=None, max_cached=2):
    """Put graph replay behind `model(data_dict)` / `loss.backward()` of a train-mode ScanQAHotPath (use_blip=True).
    Returns the model (the runner is attached outside the module tree: state_dict keys do not change).
    optimizer: also wrap_optimizer(model, optimizer).  comm_dtype / process_group: the gradient exchange under
    torch.distributed with more than one rank (fp32 on the wire = what the reference's DDP reduces in).
    max_cached: captured graph sets kept per input signature (token length, batch size of a last partial batch): a change
    of signature switches sets instead of re-capturing."""
    from torch.nn.parallel import DistributedDataParallel
    if isinstance(model, (DistributedDataParallel, torch.nn.DataParallel)):
        raise TypeError("graphed.enable: the model is wrapped in DistributedDataParallel -- the replayed backward runs no "
                        "AccumulateGrad node, so DDP's reducer would see every parameter as unused and the replicas would "
                        "diverge without an error.  Call graphed.enable(model) on the plain module INSTEAD of the DDP wrap: "
                        "under an initialised process group it exchanges the gradients itself (INTEGRATION.md section 3a).")
    object.__setattr__(model, "_graphed", GraphedRunner(model, warmup=warmup, comm_dtype=comm_dtype,
                                                        process_group=process_group, max_cached=max_cached))
    if optimizer is not None:
        wrap_optimizer(model, optimizer)
    return model


def wrap_optimizer(model, optimizer):
    """`optimizer.step()` of the unchanged loop as ONE graph replay (lib/solver.py:407-411: clip_grad_value_ +
    optimizer.step(); optim.FusedAdamW(grad_clip_value=...) has the clip inside the update kernel), and
    `optimizer.zero_grad()` without launches.  Only for optimizers whose step is capture-safe -- optim.FusedAdamW (step
    count on the device, lr / weight decay re-read from a pinned table at every replay) or a torch optimizer constructed
    with capturable=True; anything else is returned untouched.  The first wrapped step after a capture runs eagerly (the
    optimizer state must exist before a graph can be recorded), the second records, later ones replay; a step that does
    not follow a replayed backward (eval, an eager fallback) calls the optimizer itself."""
    from .optim import FusedAdamW
    runner = getattr(model, "_graphed", None)
    capturable = isinstance(optimizer, FusedAdamW) or all(g.get("capturable", False) for g in optimizer.param_groups)
    if runner is None or not capturable or getattr(optimizer, "_bq_graphed", None) is not None:
        return optimizer
    orig_step, orig_zero = optimizer.step, optimizer.zero_grad

    # (installed as BOUND methods: torch's LR schedulers patch `optimizer.step` through its `__func__` -- lib/solver.py:245-266
    # builds StepLR / MultiStepLR / CosineAnnealingLR after the optimizer, i.e. after this wrap)
    def step(self_, closure=None, **kw):
        if closure is not None or kw or not runner.replayed_backward_pending() or getattr(model, "_graphed", None) is not runner:
            return orig_step(closure, **kw) if (closure is not None or kw) else orig_step()
        return runner.optimizer_step(self_, orig_step)

    def zero_grad(self_, set_to_none=True):
        # Only while a REPLAYED forward is waiting for its backward may the static gradients stay as they are: that backward
        # overwrites every one of them (first write is an assignment).  Anywhere else -- the runner disabled, a forward that
        # took the eager path, a zero_grad() at the top of the loop -- autograd would ACCUMULATE onto what the static buffers
        # still hold, so torch's own zero_grad runs (p.grad = None; the next replayed backward points p.grad back).
        if getattr(model, "_graphed", None) is not runner or runner.graphs is None or runner._bwd_done:
            return orig_zero(set_to_none=set_to_none)
        static = runner.static_grad_ids()
        for g in self_.param_groups:
            for p in g["params"]:
                if id(p) not in static and p.grad is not None:
                    if set_to_none:
                        p.grad = None
                    else:
                        p.grad.zero_()
    import types
    optimizer.step, optimizer.zero_grad = types.MethodType(step, optimizer), types.MethodType(zero_grad, optimizer)
    optimizer._bq_graphed = (orig_step, orig_zero)
    runner._wrapped_opts.append(optimizer)
    return optimizer


def unwrap_optimizer(optimizer):
    """undo wrap_optimizer (disable(model) does it for every optimizer wrapped for that model)"""
    orig = getattr(optimizer, "_bq_graphed", None)
    if orig is not None:
        optimizer.step, optimizer.zero_grad = orig
        optimizer._bq_graphed = None
    return optimizer


class prefetch_loader(object):
    """`for data_dict in graphed.prefetch_loader(model, dataloader):` -- the loop body stays the reference's
    (lib/solver.py:475-545).  Looks ONE batch ahead and announces its point clouds to the runner, which computes their
    sampling / grouping indices on the detector stream under the fusion of the step that is about to run; every yielded
    dict carries a string key (`"_bq_geometry_key"`; the solver's `.cuda()` loop leaves strings alone, :480-483) by which
    the forward recognises the batch the prefetched indices belong to.  Attribute access (`.dataset`, `len()`) falls
    through to the wrapped loader."""

    def __init__(self, model, loader):
        self._model, self._loader, self._epoch = model, loader, 0

    def __len__(self):
        return len(self._loader)

    def __getattr__(self, name):
        return getattr(self._loader, name)

    def __iter__(self):
        self._epoch += 1
        it = iter(self._loader)
        try:
            cur = next(it)
        except StopIteration:
            return
        n = 0
        while True:
            nxt = next(it, None)
            runner = getattr(self._model, "_graphed", None)
            if isinstance(cur, dict):
                cur["_bq_geometry_key"] = "%d:%d:%d" % (id(self), self._epoch, n)
                if runner is not None and isinstance(nxt, dict) and torch.is_tensor(nxt.get("point_clouds")):
                    runner.prefetch(nxt["point_clouds"], key="%d:%d:%d" % (id(self), self._epoch, n + 1))
            yield cur
            if nxt is None:
                return
            cur, n = nxt, n + 1


def wrap_loss(model, fn):
    """The caller's loss function with graph replay behind it: `get_loss = graphed.wrap_loss(model, get_loss)` where the
    solver imports it (lib/solver.py:520-533 calls get_loss(data_dict, config, ...) right after the forward).  A call whose
    first argument is a data_dict that model() just produced by replay runs `fn` ONCE under capture -- forward of the loss
    and, behind the returned tensors' backward, the loss' own backward in front of the model's backward graphs -- and
    replays afterwards: no eager launch is left between the forward and the backward graphs (a big graph launch returns
    only when the GPU has nearly drained it, so eager launches behind it cannot be queued ahead: the ~600 small kernels of
    the detection loss and its autograd cost 8 ms of exposed launch latency per c3 step).  Anything else -- eval, another
    model's dict, changed arguments -- calls `fn` itself.  `fn` must be a pure function of the dict's tensors (plus
    constant arguments); tensors the caller put into the dict (labels) are copied into static buffers at every call."""
    def wrapped(data_dict, *args, **kw):
        runner = getattr(model, "_graphed", None)
        if runner is None or not isinstance(data_dict, dict) or data_dict.get("_bq_graphed_step") != (id(runner), runner.step_id):
            return fn(data_dict, *args, **kw)
        return runner.loss(fn, data_dict, args, kw)
    wrapped.__wrapped__ = fn
    return wrapped


def disable(model):
    runner = getattr(model, "_graphed", None)
    if runner is not None:
        for opt in runner._wrapped_opts:
            unwrap_optimizer(opt)
        runner._wrapped_opts = []
        object.__setattr__(model, "_graphed", None)
    return model


class _Bridge(torch.autograd.Function):
    """The one autograd node between the caller's loss and the captured graphs: forward hands out aliases of the graphs'
    static outputs, backward feeds the static gradient buffers and replays the backward graphs."""

    @staticmethod
    def forward(ctx, anchor, runner, *outs):
        ctx.runner, ctx.step_id = runner, runner.step_id
        ctx.set_materialize_grads(False)
        return tuple(o.view_as(o) for o in outs)

    @staticmethod
    def backward(ctx, *grads):
        ctx.runner.run_backward(ctx.step_id, grads)
        return (None, None) + (None,) * len(grads)


class _LossBridge(torch.autograd.Function):
    """backward of a replayed loss: the caller's seed (1 for loss.backward()) goes into the static seed, then the loss'
    captured backward and the model's backward graphs replay"""

    @staticmethod
    def forward(ctx, anchor, runner, key, *outs):
        ctx.runner, ctx.step_id, ctx.key = runner, runner.step_id, key
        ctx.set_materialize_grads(False)
        return tuple(o.view_as(o) for o in outs)

    @staticmethod
    def backward(ctx, *grads):
        ctx.runner.run_loss_backward(ctx.step_id, ctx.key, grads)
        return (None, None, None) + (None,) * len(grads)


def _flatten(x, out, path=()):
    """tensors of a nested tuple / list / dict structure, with their paths"""
    if torch.is_tensor(x):
        out.append((path, x))
    elif isinstance(x, dict):
        for k, v in x.items():
            _flatten(v, out, path + (("d", k),))
    elif isinstance(x, (tuple, list)):
        for i, v in enumerate(x):
            _flatten(v, out, path + (("s", i),))
    return out


def _rebuild(x, repl, path=()):
    if torch.is_tensor(x):
        return repl[path]
    if isinstance(x, dict):
        return {k: _rebuild(v, repl, path + (("d", k),)) for k, v in x.items()}
    if isinstance(x, tuple):
        return tuple(_rebuild(v, repl, path + (("s", i),)) for i, v in enumerate(x))
    if isinstance(x, list):
        return [_rebuild(v, repl, path + (("s", i),)) for i, v in enumerate(x)]
    return x


class GraphedRunner(object):
    # what one capture owns (a change of input signature stashes the set and restores / captures another: `_cache`)
    _BUNDLE = ("graphs", "sig", "static_in", "diff", "static_grads", "_st", "static_param_grads", "out_keys", "losses", "_pools",
               "anchor", "_geo_next", "_geo_cur", "next_xyz", "_geo_key", "_t_generation", "opt_graphs", "_static_ids")

    def __init__(self, model, warmup=2, comm_dtype=torch.float32, process_group=None, max_cached=2):
        self.model, self.warmup = model, int(warmup)
        self.graphs = None
        self.sig = None
        self.step_id = 0
        self._bwd_done = True
        self._opt_pending = False
        self.losses = {}      # wrap_loss: signature -> captured loss (graphs, static label buffers, result template)
        self.host_times = None   # set to {} to record the host time of every graph launch (ms, per graph)
        self.phase_events = None # set to {} to bracket every graph with events on its stream (phase_gpu_ms())
        # geometry prefetch (prefetch() / prefetch_loader): off until the first announcement
        self.prefetching = False
        self._geo_next = self._geo_cur = self.next_xyz = None
        self._geo_key = None        # which batch the indices in `_geo_next` belong to
        self._announced = None      # (point clouds, key) of the batch to compute under the NEXT forward's fusion
        self._t_generation = None
        self.opt_graphs = {}
        self._static_ids = frozenset()
        # signature -> stashed capture (LRU); `captures` counts them (a loader with varying token length re-captures on
        # every new length: pad to a fixed length, or raise max_cached)
        import collections
        self._cache, self.max_cached, self.captures = collections.OrderedDict(), max(1, int(max_cached)), 0
        # data parallel: the replayed backward fires no DDP hook -- the runner exchanges the static gradients itself
        self.comm_dtype, self.process_group = comm_dtype, process_group
        self.reducers, self.broadcaster, self.force_comm = None, None, False
        self.s_comm, self._comm_events = None, []
        self._streams_ready = False
        self._in_capture = False    # warm-up passes of a (re-)capture: no collective may be issued (see _capture)
        self._wrapped_opts = []     # optimizers wrap_optimizer changed (disable() restores them)

    # ---- what can be replayed --------------------------------------------------------------------------------------
    def usable(self, data_dict):
        m = self.model
        return (m.training and torch.is_grad_enabled() and getattr(m, "use_blip", False) and "images" in data_dict
                and torch.is_tensor(data_dict.get("point_clouds")) and data_dict["point_clouds"].is_cuda
                and isinstance(data_dict.get("question"), dict) and isinstance(data_dict.get("answer"), dict)
                and data_dict.get("phase", "train") == "train" and ops.compute_dtype() == torch.bfloat16)

    def _signature(self, data_dict):
        sig = []
        for k in _INPUT_KEYS:
            v = data_dict[k]
            items = sorted(v.items()) if isinstance(v, dict) else [("", v)]
            for kk, t in items:
                if torch.is_tensor(t):
                    sig.append((k, kk, tuple(t.shape), t.dtype, str(t.device)))
        bn = tuple(mod.momentum for mod in self.model.modules() if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm))
        return tuple(sig) + (bn,)

    def _copy_inputs(self, data_dict):
        for k in _INPUT_KEYS:
            src, dst = data_dict[k], self.static_in[k]
            if isinstance(src, dict):
                for kk, t in src.items():
                    if torch.is_tensor(t) and t.data_ptr() != dst[kk].data_ptr():
                        dst[kk].copy_(t, non_blocking=True)
            elif src.data_ptr() != dst.data_ptr():
                dst.copy_(src, non_blocking=True)

    # ---- the phases (each on one stream; the same cuts as pipeline.PhasedTrainStep) ----------------------------------
    def _inputs(self):
        dd = {k: (dict(v) if isinstance(v, dict) else v) for k, v in self.static_in.items()}
        dd["phase"] = "train"
        return dd

    def _image_fwd(self, st):
        ops.new_step(self.dev)
        st["img"] = self.model.encode_image(self._inputs())

    def _t_refresh(self, st):
        """the K-contiguous weight copies of this step's input-gradient GEMMs (fusion_state.transposed_shadow),
        re-transposed from what the caller's optimizer.step() left: detector stream, behind the detector forward, beside the
        fusion's forward chain (pipeline.PhasedTrainStep's t_refresh phase); the fusion backward waits for it"""
        if ops.TRANSPOSED_DX[0] and not (torch.cuda.is_current_stream_capturing() and ops._T_STATE["dirty"]):
            ops.refresh_transposed(self.dev)

    def _geometry(self, st):
        """sampling / grouping indices of the point clouds in `next_xyz` into the persistent `_geo_next` set
        (Pointnet2Backbone.precompute_geometry: FPS picks, centres, ball-query groups of the four SA levels, three-NN of
        the two FP levels -- coordinates only)"""
        from .pointnet2_utils import background_geometry
        with background_geometry():   # (the gentle ball-query grid: this phase runs beside the fusion chain)
            geo = self.model.detection_backbone.precompute_geometry(self.next_xyz)
        if self._geo_next is None:
            self._geo_next = {k: v.clone() for k, v in geo.items()}
        else:
            for k, v in geo.items():
                self._geo_next[k].copy_(v)

    def _det_fwd(self, st):
        dd = self._inputs()
        # the token-only head of the fusion (question / answer embeddings, BLIP_VQA3D.prepare_text) rides on the detector
        # stream, beside the image encoder; its backward opens the detector's backward phase (as pipeline.PhasedTrainStep's
        # text_prep / text_prep_bwd phases)
        bm = self.model.blip_model
        st["prep"] = bm.prepare_text(dd["question"], dd.get("answer"), self.dev) if hasattr(bm, "prepare_text") else None
        if self.prefetching:
            # next -> cur: this step's backward keeps reading `cur` while the prefetch under its fusion refills `next`
            if self._geo_cur is None:
                self._geo_cur = {k: v.clone() for k, v in self._geo_next.items()}
            else:
                for k, v in self._geo_next.items():
                    self._geo_cur[k].copy_(v)
            dd["geometry"] = self._geo_cur
        st["dd"] = self.model.detect_objects(dd)

    def _fusion_fwd(self, st):
        st["img_leaf"] = st["img"].detach().requires_grad_(True)
        st["obj_leaf"] = st["dd"]["object_feat"].detach().requires_grad_(True)
        dd = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in st["dd"].items()}
        prep = None
        if st.get("prep") is not None:
            prep = dict(st["prep"])
            st["prep_leaves"] = {k: prep[k].detach().requires_grad_(True) for k in ("q_embeds", "a_embeds") if k in prep}
            prep.update(st["prep_leaves"])
        st["fd"] = self.model.fuse(dd, st["img_leaf"], st["obj_leaf"], text_prep=prep)

    def _diff_outputs(self, st):
        """(key, tensor, phase) of every output a loss can differentiate: the detector's own outputs and what the fusion
        added (its copies of the detector's tensors are detached: their gradient belongs to the detector phase)"""
        out = []
        for k, v in st["dd"].items():
            if torch.is_tensor(v) and v.requires_grad and v.is_floating_point() and k != "object_feat":
                out.append((k, v, "det"))
        for k, v in st["fd"].items():
            if torch.is_tensor(v) and v.requires_grad and v.is_floating_point() and k not in st["dd"]:
                out.append((k, v, "fusion"))
        return out

    def _fusion_bwd(self, st):
        outs = [(t, g) for (k, t, ph), g in zip(self.diff, self.static_grads) if ph == "fusion"]
        ops.begin_deferred_wgrad()
        try:
            torch.autograd.backward([t for t, _ in outs], [g for _, g in outs])
        finally:
            ops.flush_deferred_wgrad()
        st["img_grad"], st["obj_grad"] = st["img_leaf"].grad, st["obj_leaf"].grad

    def _det_bwd(self, st):
        if st.get("prep") is not None:
            roots = [(st["prep"][k], leaf.grad) for k, leaf in st["prep_leaves"].items() if leaf.grad is not None]
            if roots:
                torch.autograd.backward([t for t, _ in roots], [g for _, g in roots])
        outs = [(t, g) for (k, t, ph), g in zip(self.diff, self.static_grads) if ph == "det"]
        torch.autograd.backward([t for t, _ in outs] + [st["dd"]["object_feat"]], [g for _, g in outs] + [st["obj_grad"]])

    def _image_bwd(self, st):
        ops.begin_deferred_wgrad()
        try:
            st["img"].backward(st["img_grad"])
        finally:
            ops.flush_deferred_wgrad()

    # ---- scheduling ---------------------------------------------------------------------------------------------------
    def _forward_phases(self, run):
        """Host ORDER of the launches matters: hipGraphLaunch returns only when the graph's packets fit into the stream's
        hardware queue, i.e. a long graph blocks the host until the GPU has consumed most of it (measured: with the
        detector forward -- ~490 nodes with its sampling chain -- launched first, the image encoder's graph reached its
        queue 8.5 ms later and the two ran back to back).  The shorter graph of each concurrent pair goes first."""
        cur = torch.cuda.current_stream(self.dev)
        if self.broadcaster is not None and not self._in_capture:
            self.broadcaster.broadcast()   # DDP's broadcast_buffers=True: rank 0's BatchNorm statistics before the forward
        for s_ in (self.s_main, self.s_det):
            s_.wait_stream(cur)
        with torch.cuda.stream(self.s_main):
            run("image_fwd")
        with torch.cuda.stream(self.s_det):
            if self.prefetching and not self._geo_ready:
                # nobody announced THIS batch a step ahead (first step, an epoch's first batch, a loop without
                # prefetch_loader): its indices are computed here, in front of the detector forward
                self.next_xyz.copy_(self.static_in["point_clouds"][..., :3])
                run("geometry")
            run("det_fwd")
            self.e_det_fwd.record(self.s_det)
            if ops.TRANSPOSED_DX[0]:
                run("t_refresh")             # (0.4 ms; the fusion backward waits for it: in FRONT of the long prefetch)
            self.e_t_refresh.record(self.s_det)
            if self.prefetching and self._announced is not None:
                nxt, key = self._announced
                self._announced = None
                self.next_xyz.copy_(nxt[..., :3], non_blocking=True)
                run("geometry")              # the NEXT batch's indices, under this step's fusion
                self._geo_key = key
            elif self.prefetching:
                self._geo_key = None         # `_geo_next` still holds this batch's indices: nothing announced
        with torch.cuda.stream(self.s_main):
            self.s_main.wait_event(self.e_det_fwd)
            run("fusion_fwd")
            self.e_fwd.record(self.s_main)
        cur.wait_event(self.e_fwd)

    def _backward_phases(self, run, grads, seeded=False):
        """grads: the caller's gradients of the differentiable outputs (None entries = unused); None = the static
        gradient buffers already hold what the backward starts from (warm-up; seeded: a replayed loss wrote them on the
        main phase stream)"""
        cur = torch.cuda.current_stream(self.dev)
        if not seeded:
            self.s_main.wait_stream(cur)
        with torch.cuda.stream(self.s_main):
            if grads is not None:
                for g, buf in zip(grads, self.static_grads):
                    if g is None:
                        buf.zero_()
                    else:
                        buf.copy_(g, non_blocking=True)
            self.e_grads.record(self.s_main)
            self.s_main.wait_event(self.e_t_refresh)
            run("fusion_bwd")
            self.e_fused.record(self.s_main)
            run("image_bwd")      # (before the detector's: see _forward_phases)
            self.e_img_bwd.record(self.s_main)
        self.s_det.wait_event(self.e_grads)
        self.s_det.wait_event(self.e_fused)
        with torch.cuda.stream(self.s_det):
            run("det_bwd")
            self.e_det_bwd.record(self.s_det)
        if self._probe is not None:
            self._probe("end")
        cur.wait_event(self.e_img_bwd)
        cur.wait_event(self.e_det_bwd)
        if self.reducers and not self._in_capture:
            # (never during a re-capture's warm-up passes: under padding='longest' the token shapes -- part of the capture
            # signature -- differ per rank, so one rank can re-capture while the others replay; its warm-up gradients are
            # seeded with 1e-3 and its collectives would pair up with the other ranks' REAL ones, one step apart.  With the
            # warm-up silent every rank issues the same collectives per step whatever it captures.)
            # data parallel: the fusion group (3/4 of the bytes: gradients complete when fusion_bwd ends) travels on the
            # communication stream under the image / detector backward, the rest when both have finished
            sc = self.s_comm
            sc.wait_event(self.e_fused)
            with torch.cuda.stream(sc):
                if "fusion" in self.reducers:
                    self.reducers["fusion"].all_reduce()
                sc.wait_event(self.e_img_bwd)
                sc.wait_event(self.e_det_bwd)
                if "rest" in self.reducers:
                    self.reducers["rest"].all_reduce()
                self.e_comm.record(sc)
            cur.wait_event(self.e_comm)

    _probe = None   # capture(): called with the phase name after every eager phase of the last warm-up pass

    def _eager(self, st):
        def run(name):
            getattr(self, "_" + name)(st)
            if self._probe is not None:
                self._probe(name)
        return run

    def _replay(self, name):
        if self.host_times is None and self.phase_events is None:
            self.graphs[name].replay()
            return
        import time
        s_ = torch.cuda.current_stream(self.dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s_)
        t0 = time.perf_counter()
        self.graphs[name].replay()
        if self.host_times is not None:
            self.host_times.setdefault(name, []).append((time.perf_counter() - t0) * 1e3)
        e1.record(s_)
        if self.phase_events is not None:
            self.phase_events.setdefault(name, []).append((e0, e1))

    def phase_gpu_ms(self):
        """after a synchronize: {graph: (mean start, mean end)} in ms from the start of the step's first graph"""
        out = {}
        steps = min(len(v) for v in self.phase_events.values())
        ref = self.phase_events["image_fwd"][-steps:]
        for name, evs in self.phase_events.items():
            evs = evs[-steps:]     # (aligned at the END: a step that computed its own geometry has an extra, earlier entry)
            a = [ref[i][0].elapsed_time(evs[i][0]) for i in range(steps)]
            b = [ref[i][0].elapsed_time(evs[i][1]) for i in range(steps)]
            out[name] = (sum(a) / steps, sum(b) / steps)
        return out

    # ---- capture --------------------------------------------------------------------------------------------------------
    def capture(self, data_dict):
        # single-stream graphs: a graph with an internal fork (fusion_ops.fork: the decoder's hoisted K/V projection on a
        # side stream) is enqueued node by node by this runtime -- 8.8 ms of host time for the fusion forward, which then
        # ran 12.2 ms on the GPU instead of ~4.5
        prev = ops.set_overlap(False)
        announced, self._announced = self._announced, None   # (the warm-up passes must not consume the announcement)
        self._in_capture = True
        try:
            self._capture(data_dict)
        finally:
            self._in_capture = False
            ops.set_overlap(prev)
            self._announced = announced

    def _capture(self, data_dict):
        m = self.model
        # an earlier eager forward can survive as a reference CYCLE (fusion_ops.HoistedKV <-> its autograd node): its
        # AccumulateGrad nodes -- born on the caller's stream -- would be reused by the capture and hipStreamEndCapture faults
        # on them; collect before capturing
        import gc
        gc.collect()
        self.dev = dev = data_dict["point_clouds"].device
        self.static_in = {k: ({kk: (t.clone() if torch.is_tensor(t) else t) for kk, t in v.items()} if isinstance(v, dict)
                              else v.clone()) for k, v in ((k, data_dict[k]) for k in _INPUT_KEYS)}
        if not self._streams_ready:   # (streams and events are shared by every captured set of this runner)
            self.s_main, self.s_det = ops.phase_streams(dev, -1, 0)   # (the process-wide pair, shared with PhasedTrainStep)
            (self.e_det_fwd, self.e_fwd, self.e_grads, self.e_fused, self.e_det_bwd, self.e_img_bwd, self.e_loss,
             self.e_t_refresh, self.e_opt, self.e_comm) = (torch.cuda.Event() for _ in range(10))
            self._streams_ready = True
            self._setup_data_parallel(dev)
        self.anchor = torch.zeros(1, device=dev, requires_grad=True)
        self._geo_next = self._geo_cur = None
        self._geo_key, self._geo_ready = None, False
        self.next_xyz = data_dict["point_clouds"][..., :3].contiguous().clone() if self.prefetching else None
        self.opt_graphs = {}
        params = [p for p in m.parameters() if p.requires_grad]
        # warm-up: eager steps on the phase streams (autograd's AccumulateGrad nodes remember the stream they were born on;
        # kernels compile / caches fill), seeded with ones; buffers (BatchNorm statistics) and gradients put back afterwards
        torch.cuda.synchronize(dev)
        # BUFFERS only (BatchNorm statistics, num_batches_tracked).  Round 4 filtered m.state_dict() by isinstance(Parameter):
        # state_dict() hands out DETACHED tensors, so every parameter was "restored" too -- a copy_ that bumps its version
        # counter, which made every bf16 operand copy look stale at capture time: the captured graphs re-cast ~250 weights per
        # replay (1.3 ms of multi-tensor copies + 188 casts in the fusion forward alone) and their input-gradient GEMMs fell
        # back to the contraction-major weight read (+1.6 ms in the image backward)
        saved_buf = {k: v.detach().clone() for k, v in m.named_buffers()}
        saved_grad = [p.grad for p in params]
        for p in params:
            p.grad = None
        st = {}
        seen = {}     # data parallel: which gradients exist when the fusion backward ends, and which of them a LATER phase
        #               accumulates into (the token embeddings: text_prep's backward runs inside det_bwd)

        def probe(name):
            if name == "fusion_bwd":
                seen["fusion"] = {id(p): (id(p.grad), p.grad._version) for p in params if p.grad is not None}
                seen["keep"] = [p.grad for p in params if p.grad is not None]   # (ids stay unique while these live)
            elif name == "end":
                seen["late"] = {id(p) for p in params if p.grad is not None
                                and seen["fusion"].get(id(p)) != (id(p.grad), p.grad._version)}
                seen.pop("keep", None)
                seen["all"] = [p for p in params if p.grad is not None]
        n_warm = max(1, self.warmup)
        for it in range(n_warm):
            st = {}
            self._probe = probe if it == n_warm - 1 else None
            self._forward_phases(self._eager(st))
            self.diff = self._diff_outputs(st)
            self.static_grads = [torch.ones_like(t) * 1e-3 for _, t, _ in self.diff]
            self._backward_phases(self._eager(st), None)
            self._probe = None
            for p in params:
                p.grad = None
        torch.cuda.synchronize(dev)
        with torch.no_grad():
            for k, v in m.named_buffers():
                v.copy_(saved_buf[k])
        if ops.TRANSPOSED_DX[0]:
            ops.refresh_transposed(dev)   # (builds the device table of what the warm-up registered: not possible inside a capture)
            self._t_generation = ops.transposed_generation()   # (its tables and operands live as long as these graphs)
        # capture: one graph per phase, a pool per stream (the two streams' graphs run concurrently); the geometry and the
        # weight-copy refresh in pools of their own (they replay out of capture order: before OR after the detector forward)
        pools = self._pools = {"main": torch.cuda.graph_pool_handle(), "det": torch.cuda.graph_pool_handle(),
                               "geo": torch.cuda.graph_pool_handle(), "aux": torch.cuda.graph_pool_handle()}
        self.losses = {}
        order = (("image_fwd", "main", "main"),) + ((("geometry", "det", "geo"),) if self.prefetching else ()) + (
            ("det_fwd", "det", "det"),) + ((("t_refresh", "det", "aux"),) if ops.TRANSPOSED_DX[0] else ()) + (
            ("fusion_fwd", "main", "main"),
            ("fusion_bwd", "main", "main"), ("image_bwd", "main", "main"), ("det_bwd", "det", "det"))
        st, graphs = {}, {}
        streams = {"main": self.s_main, "det": self.s_det}
        for name, which, pool in order:
            if name == "fusion_bwd":
                self.diff = self._diff_outputs(st)
                self.static_grads = [torch.zeros_like(t) for _, t, _ in self.diff]
            g = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(g, pool=pools[pool], stream=streams[which]):
                    getattr(self, "_" + name)(st)
            except Exception as e:
                raise RuntimeError("graphed.enable: phase '%s' could not be captured: %s" % (name, e)) from e
            graphs[name] = g
            torch.cuda.synchronize(dev)
        self._st = st                       # keeps the captured autograd graph and the static outputs alive
        self.graphs = graphs
        self.static_param_grads = [(p, p.grad) for p in params if p.grad is not None]
        self._static_ids = frozenset(id(p) for p, _ in self.static_param_grads)
        for p, g in zip(params, saved_grad):
            p.grad = g
        self.out_keys = [k for k, _, _ in self.diff]
        self.sig = self._signature(data_dict)
        self.captures += 1
        if self.captures == 4:
            import warnings
            warnings.warn("graphed: %d captures so far -- every new input signature (token length under padding='longest', a "
                          "last partial batch, a BatchNorm-momentum step) costs warm-up steps, eight captures and a memory pool "
                          "of its own; pad the tokens to a fixed length / drop_last, or raise enable(max_cached=)" % self.captures)
        if self._comm_on() and self.reducers is None:
            self._build_reducers(seen)
        elif self.reducers:
            covered = {id(p) for r in self.reducers.values() for p in r.params}
            if covered != set(self._static_ids):
                raise RuntimeError("graphed: this capture produces gradients for a different parameter set than the one the "
                                   "gradient exchange was built for")

    # ---- data parallel ------------------------------------------------------------------------------------------------
    def _comm_on(self):
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized() and (dist.get_world_size(self.process_group) > 1 or self.force_comm)

    def _setup_data_parallel(self, dev):
        """what DDP's constructor does once (scripts/train.py:346-347): every replica starts from rank 0's parameters and
        buffers; and its broadcast_buffers=True: rank 0's buffers before every forward"""
        if not self._comm_on():
            return
        from . import ddp
        ddp.broadcast_parameters(self.model, 0, self.process_group)
        ops.refresh_shadows(only_with_grad=False)
        self.broadcaster = ddp.BufferBroadcaster(self.model, 0, self.process_group)
        self.broadcaster.force = self.force_comm
        self.s_comm = torch.cuda.Stream(device=dev)

    def _build_reducers(self, seen):
        from . import ddp
        fusion = [p for p in seen["all"] if id(p) in seen["fusion"] and id(p) not in seen["late"]]
        fid = {id(p) for p in fusion}
        rest = [p for p in seen["all"] if id(p) not in fid]
        missing = set(self._static_ids) ^ {id(p) for p in seen["all"]}
        if missing:
            raise RuntimeError("graphed: the warm-up pass and the captured pass disagree on which parameters receive gradients")
        self.reducers = {}
        for name, ps in (("fusion", fusion), ("rest", rest)):
            if ps:
                r = ddp.PackedGradReducer(ps, comm_dtype=self.comm_dtype, process_group=self.process_group)
                r.force = self.force_comm
                self.reducers[name] = r

    # ---- capture cache ------------------------------------------------------------------------------------------------
    def _stash(self):
        if self.graphs is not None and self.sig is not None:
            self._cache[self.sig] = {k: getattr(self, k, None) for k in self._BUNDLE}
            self._cache.move_to_end(self.sig)
            while len(self._cache) > self.max_cached:
                self._cache.popitem(last=False)

    def _switch(self, data_dict):
        """the captured set for this input signature: the current one, a cached one, or a new capture"""
        sig = self._signature(data_dict)
        if self.graphs is not None and sig == self.sig:
            return
        self._stash()
        hit = self._cache.pop(sig, None)
        if hit is not None and (hit.get("next_xyz") is not None) == self.prefetching:
            for k, v in hit.items():
                setattr(self, k, v)
            self._geo_key = None          # (its prefetched indices are from another time)
            return
        self.graphs = None
        self.capture(data_dict)

    # ---- the two calls ------------------------------------------------------------------------------------------------
    def forward(self, data_dict):
        self._switch(data_dict)
        if not self._bwd_done:
            pass   # (a forward without a backward -- e.g. a skipped step -- is fine: the next replay overwrites everything)
        self.step_id += 1
        self._bwd_done = False
        self._opt_pending = False
        if self.prefetching:
            # do the indices in `_geo_next` belong to THIS batch?  prefetch_loader's string key, or the very tensor that
            # was announced (same object, not written since)
            pc = data_dict["point_clouds"]
            key = data_dict.get("_bq_geometry_key")
            # (the announced tensor itself is kept in the key and compared by identity: an id() can be recycled for another
            # batch tensor once the announced one has died)
            gk = self._geo_key
            self._geo_ready = gk is not None and (
                (key is not None and key == gk)
                or (isinstance(gk, tuple) and len(gk) == 3 and gk[0] == "tensor" and gk[1] is pc and gk[2] == pc._version))
        self._copy_inputs(data_dict)
        self._forward_phases(self._replay)
        # (DETACHED inputs: an edge into the captured autograd graph would make this backward walk it eagerly)
        outs = _Bridge.apply(self.anchor, self, *[t.detach() for _, t, _ in self.diff])
        dd = dict(data_dict)
        for k in ("point_clouds", "images", "question", "answer"):
            dd[k] = data_dict[k]
        # everything the eager forward would have put into the dict: non-differentiable entries as the static tensors
        # (detached), differentiable ones through the bridge node
        for src in (self._st["dd"], self._st["fd"]):
            for k, v in src.items():
                if k in _INPUT_KEYS or k == "phase":
                    continue
                dd[k] = v.detach() if torch.is_tensor(v) else v
        for k, o in zip(self.out_keys, outs):
            dd[k] = o
        if "blip_loss" in dd:
            dd["decoder_loss"] = dd["blip_loss"]
        dd["_bq_graphed_step"] = (id(self), self.step_id)   # (wrap_loss recognises the dict of THIS replay)
        return dd

    # ---- the caller's loss under replay (wrap_loss) -----------------------------------------------------------------------
    def _loss_signature(self, fn, data_dict, args, kw):
        own = set(self._st["dd"]) | set(self._st["fd"]) | {"_bq_graphed_step"}
        ext = tuple(sorted((k, tuple(v.shape), v.dtype) for k, v in data_dict.items() if k not in own and torch.is_tensor(v)))
        # host constants the captured loss may branch on ("phase", flags) are part of the signature; entries that change
        # every step and that no loss reads -- the solver's data_dict["iteration"] (lib/solver.py:488), this module's own
        # "_bq_*" keys -- must not be: each new value would re-capture the loss (two eager passes + two captures per step)
        const = tuple(sorted((k, repr(v)) for k, v in data_dict.items()
                             if k not in own and not torch.is_tensor(v) and not isinstance(v, dict)
                             and not (isinstance(k, str) and (k.startswith("_bq_") or k in VOLATILE_KEYS))))
        return (id(fn), ext, const, repr(args), repr(sorted(kw.items())))

    def _capture_loss(self, fn, data_dict, args, kw):
        own = set(self._st["dd"]) | set(self._st["fd"])
        ext = {k: v.clone() for k, v in data_dict.items() if k not in own and torch.is_tensor(v) and k != "_bq_graphed_step"}
        rest = {k: v for k, v in data_dict.items() if k not in own and k not in ext and k != "_bq_graphed_step"}
        diff_keys = {k: i for i, (k, _, _) in enumerate(self.diff)}
        rec = {"ext": ext}

        def build():
            """the dict fn sees: the model's static outputs -- differentiable ones as fresh leaves --, the static copies of
            the caller's tensors, the caller's constants"""
            dd, leaves = dict(rest), {}
            for src in (self._st["dd"], self._st["fd"]):
                for k, v in src.items():
                    dd[k] = v.detach() if torch.is_tensor(v) else v
            for k, i in diff_keys.items():
                leaves[k] = self.diff[i][1].detach().requires_grad_(True)
                dd[k] = leaves[k]
            if "blip_loss" in leaves:
                dd["decoder_loss"] = leaves["blip_loss"]
            dd.update(ext)
            return dd, leaves

        def fwd(state):
            dd, leaves = build()
            ret = fn(dd, *args, **kw)
            flat = _flatten(ret, [])
            state.update(ret=ret, flat=flat, leaves=leaves, roots=[(p, t) for p, t in flat if t.requires_grad and t.grad_fn is not None])

        def bwd(state):
            roots = state["roots"]
            torch.autograd.backward([t for _, t in roots], [s for s in rec["seeds"]])
            for k, i in diff_keys.items():   # what the model's backward graphs are seeded with
                g = state["leaves"][k].grad
                if g is None:
                    self.static_grads[i].zero_()
                else:
                    self.static_grads[i].copy_(g)

        # warm-up (eager, on the main phase stream), then two graphs
        cur = torch.cuda.current_stream(self.dev)
        self.s_main.wait_stream(cur)
        with torch.cuda.stream(self.s_main):
            for _ in range(2):
                state = {}
                fwd(state)
                rec["seeds"] = [torch.ones_like(t) for _, t in state["roots"]]
                bwd(state)
        torch.cuda.synchronize(self.dev)
        state = {}
        g_f, g_b = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        # (a pool of their own: the loss graphs replay BETWEEN graphs captured before them, so they must not reuse memory
        # those freed at capture, and the returned loss values must outlive the backward graphs' temporaries)
        pool = torch.cuda.graph_pool_handle()
        with torch.cuda.graph(g_f, pool=pool, stream=self.s_main):
            fwd(state)
        rec["seeds"] = [torch.ones_like(t) for _, t in state["roots"]]
        torch.cuda.synchronize(self.dev)
        with torch.cuda.graph(g_b, pool=pool, stream=self.s_main):
            bwd(state)
        torch.cuda.synchronize(self.dev)
        rec.update(state=state, g_f=g_f, g_b=g_b, root_paths=[p for p, _ in state["roots"]])
        return rec

    def loss(self, fn, data_dict, args, kw):
        key = self._loss_signature(fn, data_dict, args, kw)
        rec = self.losses.get(key)
        if rec is None:
            rec = self.losses[key] = self._capture_loss(fn, data_dict, args, kw)
        cur = torch.cuda.current_stream(self.dev)
        self.s_main.wait_stream(cur)
        with torch.cuda.stream(self.s_main):
            for k, buf in rec["ext"].items():
                t = data_dict[k]
                if t.data_ptr() != buf.data_ptr():
                    buf.copy_(t, non_blocking=True)
            self._replay_named("loss_fwd", rec["g_f"])
            self.e_loss.record(self.s_main)
        cur.wait_event(self.e_loss)
        st = rec["state"]
        roots = [t.detach() for _, t in st["roots"]]
        outs = _LossBridge.apply(self.anchor, self, key, *roots)
        repl = {p: t.detach() for p, t in st["flat"]}
        repl.update({p: o for p, o in zip(rec["root_paths"], outs)})
        ret = _rebuild(st["ret"], repl)
        # a loss that returns the dict it was given (get_loss -> (loss, data_dict)): the caller keeps using ITS dict
        flat_dicts = [ret] if isinstance(ret, dict) else [r for r in (ret if isinstance(ret, (tuple, list)) else ()) if isinstance(r, dict)]
        for d in flat_dicts:
            for k in list(d):
                if k in data_dict and k not in rec["ext"] and not any(k == p[-1][1] for p in rec["root_paths"] if p):
                    d[k] = data_dict[k]
        return ret

    def run_loss_backward(self, step_id, key, grads):
        if step_id != self.step_id:
            raise RuntimeError("graphed: backward of a stale forward (its static outputs were overwritten by a later forward)")
        if self._bwd_done:
            raise RuntimeError("graphed: one backward per forward (the captured graphs overwrite the gradients)")
        rec = self.losses[key]
        cur = torch.cuda.current_stream(self.dev)
        self.s_main.wait_stream(cur)
        with torch.cuda.stream(self.s_main):
            for g, seed in zip(grads, rec["seeds"]):
                if g is None:
                    seed.zero_()
                else:
                    seed.copy_(g, non_blocking=True)
            self._replay_named("loss_bwd", rec["g_b"])
        self._bwd_done = True
        self._point_grads()
        self._backward_phases(self._replay, None, seeded=True)
        self._opt_pending = True

    def run_backward(self, step_id, grads):
        if step_id != self.step_id:
            raise RuntimeError("graphed: backward of a stale forward (its static outputs were overwritten by a later forward)")
        if self._bwd_done:
            raise RuntimeError("graphed: one backward per forward (the captured graphs overwrite the gradients)")
        self._bwd_done = True
        self._point_grads()
        self._backward_phases(self._replay, grads)
        self._opt_pending = True

    def _point_grads(self):
        """`p.grad` = the static gradient the captured backward writes (an optimizer.zero_grad() before the backward set it
        to None); done BEFORE the replay is enqueued: the gradient exchange reads p.grad when it is issued"""
        for p, g in self.static_param_grads:
            if p.grad is not g:
                p.grad = g

    def static_grad_ids(self):
        return self._static_ids

    # ---- the caller's optimizer under replay (wrap_optimizer) -------------------------------------------------------------
    def replayed_backward_pending(self):
        return self.graphs is not None and self._bwd_done and self._opt_pending

    def optimizer_step(self, opt, orig_step):
        self._opt_pending = False
        rec = self.opt_graphs.setdefault(id(opt), {"n": 0, "g": None, "hyper": None})
        cur = torch.cuda.current_stream(self.dev)
        if not hasattr(opt, "sync_hyperparams"):
            # a torch optimizer (capturable=True) bakes every HOST-valued hyperparameter into the recorded launches (a tensor
            # lr is read on the device): an LR scheduler's decay would be ignored on replay without an error -- record again
            # when one of them has changed (StepLR / MultiStepLR: a few times per run; a per-step schedule wants a tensor lr
            # or optim.FusedAdamW, whose captured launch re-reads its table)
            hyper = tuple(tuple(sorted((k, v) for k, v in g.items() if k != "params" and isinstance(v, (int, float, bool, tuple))))
                          for g in opt.param_groups)
            if rec["g"] is not None and rec["hyper"] != hyper:
                rec["g"] = None
            rec["hyper"] = hyper
        if rec["g"] is None:
            rec["n"] += 1
            if rec["n"] < 2:
                return orig_step()            # eager: the optimizer's state (moments, device tables) comes into being
            self.s_main.wait_stream(cur)
            torch.cuda.synchronize(self.dev)
            g = torch.cuda.CUDAGraph()
            if "opt" not in self._pools:
                self._pools["opt"] = torch.cuda.graph_pool_handle()
            with torch.cuda.graph(g, pool=self._pools["opt"], stream=self.s_main):
                orig_step()
            rec["g"] = g
        if hasattr(opt, "sync_hyperparams"):
            opt.sync_hyperparams()            # LR schedulers act on param_groups; the captured launch reads the pinned table
        self.s_main.wait_stream(cur)
        with torch.cuda.stream(self.s_main):
            self._replay_named("optimizer", rec["g"])
            self.e_opt.record(self.s_main)
        cur.wait_event(self.e_opt)
        # (no Python runs in a replay: the post-step hook that marks the K-contiguous weight copies stale did not fire)
        ops.mark_transposed_stale()
        return None

    def _replay_named(self, name, g):
        if self.host_times is None and self.phase_events is None:
            g.replay()
            return
        saved, self.graphs = self.graphs, dict(self.graphs, **{name: g})
        try:
            self._replay(name)
        finally:
            self.graphs = saved

    # ---- geometry prefetch ------------------------------------------------------------------------------------------------
    def prefetch(self, next_point_clouds, key=None):
        """Announce the point clouds of the NEXT step ((B, N, 3 + C) or (B, N, 3), host or device): their sampling /
        grouping indices are computed on the detector stream behind the detector forward of the forward that is called
        next, i.e. under its fusion -- and used by the forward after that instead of being computed in front of its
        detector phase (3.7 ms of the c3 step).  key: how that later forward recognises its batch -- `data_dict
        ["_bq_geometry_key"]` (prefetch_loader sets it); default: the announced tensor object itself.  The first
        announcement switches the runner to the prefetching schedule (one re-capture: the detector forward then takes
        the indices as an input)."""
        if not torch.is_tensor(next_point_clouds) or next_point_clouds.dim() != 3 or next_point_clouds.shape[-1] < 3:
            raise ValueError("graphed.prefetch: point clouds of shape (B, N, >= 3) expected")
        if key is None:
            key = ("tensor", next_point_clouds, next_point_clouds._version)   # (a strong reference: see forward())
        if not self.prefetching:
            self.prefetching = True
            self._stash()
            self._cache.clear()
            self.graphs = None               # re-capture with the geometry as a phase of its own
        if self.next_xyz is not None and tuple(next_point_clouds.shape[:2]) != tuple(self.next_xyz.shape[:2]):
            return                            # (a batch of another shape: its forward computes its own indices)
        self._announced = (next_point_clouds, key)

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
ordinary eager path.  Without the next batch nothing can be prefetched: the sampling / grouping indices are computed
inside the detector forward (PhasedTrainStep hides them under the fusion of the previous step; `prefetch()` below offers
the same to a loop that can name its next point clouds).
"""
import torch

from . import fusion_ops as ops


_INPUT_KEYS = ("point_clouds", "images", "question", "answer")


def enable(model, warmup=2):
    """Put graph replay behind `model(data_dict)` / `loss.backward()` of a train-mode ScanQAHotPath (use_blip=True).
    Returns the model (the runner is attached outside the module tree: state_dict keys do not change)."""
    object.__setattr__(model, "_graphed", GraphedRunner(model, warmup=warmup))
    return model


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
    if getattr(model, "_graphed", None) is not None:
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
    def __init__(self, model, warmup=2):
        self.model, self.warmup = model, int(warmup)
        self.graphs = None
        self.sig = None
        self.step_id = 0
        self._bwd_done = True
        self._geo_next = None
        self.losses = {}      # wrap_loss: signature -> captured loss (graphs, static label buffers, result template)
        self.host_times = None   # set to {} to record the host time of every graph launch (ms, per graph)
        self.phase_events = None # set to {} to bracket every graph with events on its stream (phase_gpu_ms())

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
        # the K-contiguous weight copies of this step's text-side input-gradient GEMMs (fusion_state.transposed_shadow),
        # re-transposed from what the caller's optimizer.step() left: here the main phase stream idles until the detector
        # forward (the longer of the two in this loop) hands over
        if ops.TRANSPOSED_DX[0] and not (torch.cuda.is_current_stream_capturing() and ops._T_STATE["dirty"]):
            ops.refresh_transposed(self.dev)

    def _det_fwd(self, st):
        dd = self._inputs()
        # the token-only head of the fusion (question / answer embeddings, BLIP_VQA3D.prepare_text) rides on the detector
        # stream, beside the image encoder; its backward opens the detector's backward phase (as pipeline.PhasedTrainStep's
        # text_prep / text_prep_bwd phases)
        bm = self.model.blip_model
        st["prep"] = bm.prepare_text(dd["question"], dd.get("answer"), self.dev) if hasattr(bm, "prepare_text") else None
        if self._geo_next is not None:
            dd["geometry"] = self._geo_next
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
        for s_ in (self.s_main, self.s_det):
            s_.wait_stream(cur)
        with torch.cuda.stream(self.s_main):
            run("image_fwd")
        with torch.cuda.stream(self.s_det):
            run("det_fwd")
            self.e_det_fwd.record(self.s_det)
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
            run("fusion_bwd")
            self.e_fused.record(self.s_main)
            run("image_bwd")      # (before the detector's: see _forward_phases)
            self.e_img_bwd.record(self.s_main)
        self.s_det.wait_event(self.e_grads)
        self.s_det.wait_event(self.e_fused)
        with torch.cuda.stream(self.s_det):
            run("det_bwd")
            self.e_det_bwd.record(self.s_det)
        cur.wait_event(self.e_img_bwd)
        cur.wait_event(self.e_det_bwd)

    def _eager(self, st):
        return lambda name: getattr(self, "_" + name)(st)

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
        out, ref = {}, self.phase_events["image_fwd"]
        steps = min(len(v) for v in self.phase_events.values())
        for name, evs in self.phase_events.items():
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
        try:
            self._capture(data_dict)
        finally:
            ops.set_overlap(prev)

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
        self.s_main = torch.cuda.Stream(device=dev, priority=-1)
        self.s_det = torch.cuda.Stream(device=dev, priority=0)
        (self.e_det_fwd, self.e_fwd, self.e_grads, self.e_fused, self.e_det_bwd, self.e_img_bwd, self.e_loss) = (torch.cuda.Event() for _ in range(7))
        self.anchor = torch.zeros(1, device=dev, requires_grad=True)
        params = [p for p in m.parameters() if p.requires_grad]
        # warm-up: eager steps on the phase streams (autograd's AccumulateGrad nodes remember the stream they were born on;
        # kernels compile / caches fill), seeded with ones; buffers (BatchNorm statistics) and gradients put back afterwards
        torch.cuda.synchronize(dev)
        saved_buf = {k: v.detach().clone() for k, v in m.state_dict().items() if not isinstance(v, torch.nn.Parameter)}
        saved_grad = [p.grad for p in params]
        for p in params:
            p.grad = None
        st = {}
        for _ in range(max(1, self.warmup)):
            st = {}
            self._forward_phases(self._eager(st))
            self.diff = self._diff_outputs(st)
            self.static_grads = [torch.ones_like(t) * 1e-3 for _, t, _ in self.diff]
            self._backward_phases(self._eager(st), None)
            for p in params:
                p.grad = None
        torch.cuda.synchronize(dev)
        with torch.no_grad():
            for k, v in m.state_dict().items():
                if k in saved_buf:
                    v.copy_(saved_buf[k])
        if ops.TRANSPOSED_DX[0]:
            ops.refresh_transposed(dev)   # (builds the device table of what the warm-up registered: not possible inside a capture)
        # capture: one graph per phase, a pool per stream (the two streams' graphs run concurrently)
        pools = self._pools = {"main": torch.cuda.graph_pool_handle(), "det": torch.cuda.graph_pool_handle()}
        self.losses = {}
        order = (("image_fwd", "main"), ("det_fwd", "det"), ("fusion_fwd", "main"), ("fusion_bwd", "main"), ("image_bwd", "main"),
                 ("det_bwd", "det"))
        st, graphs = {}, {}
        streams = {"main": self.s_main, "det": self.s_det}
        for name, which in order:
            if name == "fusion_bwd":
                self.diff = self._diff_outputs(st)
                self.static_grads = [torch.zeros_like(t) for _, t, _ in self.diff]
            g = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(g, pool=pools[which], stream=streams[which]):
                    getattr(self, "_" + name)(st)
            except Exception as e:
                raise RuntimeError("graphed.enable: phase '%s' could not be captured: %s" % (name, e)) from e
            graphs[name] = g
            torch.cuda.synchronize(dev)
        self._st = st                       # keeps the captured autograd graph and the static outputs alive
        self.graphs = graphs
        self.static_param_grads = [(p, p.grad) for p in params if p.grad is not None]
        for p, g in zip(params, saved_grad):
            p.grad = g
        self.out_keys = [k for k, _, _ in self.diff]
        self.sig = self._signature(data_dict)

    # ---- the two calls ------------------------------------------------------------------------------------------------
    def forward(self, data_dict):
        if self.graphs is None or self._signature(data_dict) != self.sig:
            self.graphs = None
            self.capture(data_dict)
        if not self._bwd_done:
            pass   # (a forward without a backward -- e.g. a skipped step -- is fine: the next replay overwrites everything)
        self.step_id += 1
        self._bwd_done = False
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
        const = tuple(sorted((k, repr(v)) for k, v in data_dict.items()
                             if k not in own and not torch.is_tensor(v) and not isinstance(v, dict)))
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
            rec["g_f"].replay()
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
            rec["g_b"].replay()
        self._bwd_done = True
        self._backward_phases(self._replay, None, seeded=True)
        for p, g in self.static_param_grads:
            if p.grad is not g:
                p.grad = g

    def run_backward(self, step_id, grads):
        if step_id != self.step_id:
            raise RuntimeError("graphed: backward of a stale forward (its static outputs were overwritten by a later forward)")
        if self._bwd_done:
            raise RuntimeError("graphed: one backward per forward (the captured graphs overwrite the gradients)")
        self._bwd_done = True
        self._backward_phases(self._replay, grads)
        for p, g in self.static_param_grads:   # (an optimizer.zero_grad() before the backward set them to None)
            if p.grad is not g:
                p.grad = g

    def prefetch(self, next_point_clouds):
        """optional: the NEXT step's sampling / grouping indices now, on the detector stream (beside whatever runs) -- the
        following forward uses them instead of computing them in its detector phase.  Needs a re-capture the first time."""
        raise NotImplementedError("graphed.prefetch: use pipeline.PhasedTrainStep(next_batch=...) for the prefetching schedule")

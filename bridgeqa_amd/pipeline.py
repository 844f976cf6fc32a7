"""One training step of ScanQAHotPath scheduled as phases on two HIP streams.

The step has two independent heavy branches -- the image encoder (GPU-filling GEMMs) and the point-cloud detector
(FPS / ball query / small convolutions: latency-bound, a few workgroups at a time) -- joined by the twin
cross-attention fusion.  Measured on MI355X (profiles/r01_*): captured as ONE multi-stream HIP graph the two
branches barely overlap (the runtime enqueues the graph's streams one list after the other, and the detector's
kernels starve behind the encoder's on an equal-priority queue).  Here every phase is its own graph, replayed on an
explicit stream with explicit events:

    main stream:  image fwd ............. fusion fwd+bwd ..........-> image bwd ........ optimizer
    det  stream:  detector fwd -> (event) geometry of the NEXT batch -> detector bwd -> (event)

Measured and NOT adopted (round 2, c3, per-phase HIP events, BQ_PIPE_TRACE=1): moving the fusion phase's tail -- its
grouped weight-gradient GEMMs (2.3 ms) and the AdamW update of its 399 M parameters (2.0 ms) -- to a third stream under
the image / detector backward (split_fusion_tail / split_fusion_opt below).  The fusion phase hands over 1.6 ms earlier
and the final optimizer phase shrinks from 2.7 to 0.5 ms, but the image backward stretches from 17.9 to 21.8 ms and the
detector backward from 13.4 to 15.9 ms: a HBM-bound update next to GEMMs that stream their operands from HBM only moves
the time (one dX GEMM that normally takes 73 us ran 2.0 ms beside the AdamW kernel).  Step, same box: 46.5 ms inline,
46.7 ms with the weight gradients on the side stream, 46.8 ms with the optimizer there too.  Round 3 repeated the
optimizer part with the side stream confined to 16 / 32 / 64 CUs (hipExtStreamCreateWithCUMask): 40.1 - 42.3 ms against
38.6 -- the backward window is bound by the SUM of its work, memory traffic included.  The schedules and their knobs
(split_fusion_tail / split_fusion_opt, CU-masked encoder and detector streams) were removed in round 3; DESIGN.md §5
keeps the numbers.

The fusion phase is ~1200 short kernels that leave most of the 256 CUs idle, and FPS / ball query / three-NN depend
on coordinates only (no parameters): the sampling and grouping indices of the next batch are computed under it
(4 ms per step hidden, measured) and handed to the next detector forward through a copy, so the backward of the
current step still reads the indices it was built with.

What overlaps and what does not (rocprofv3 traces, profiles/r01_*): short kernels overlap short kernels (the twin
branches of the fusion; the geometry prefetch under the fusion), and a kernel that is already resident keeps its CUs
(FPS under the image encoder); but while a stream of GPU-filling kernels runs (image encoder forward / backward) the
other stream's kernels are only dispatched at its kernel boundaries, so detector and encoder phases execute almost
back to back in launch order whatever the priorities -- CU masking would fix that at the price of taking the CUs
away from the encoder for the whole step.  The schedule therefore puts the detector's parameter-free work where it
overlaps and accepts the rest.

Autograd is cut at the two tensors that cross streams (image_embeds, object_feat): the fusion phase differentiates
w.r.t. detached leaves and the branch backward phases are seeded with those leaf gradients -- the same gradients as
one backward over the whole graph (chain rule), asserted in tests/test_pipeline_*.

Reference: the work of one iteration of scripts/train.py -> lib/solver.py:_forward/_backward (models/qa_module.py
forward) restricted to the hot path; the optimizer is the caller's.
"""
import torch

from . import fusion_ops as ops


_DET_LOSS_LATE = [True]   # the fusion waits for the detector's outputs only (False: also for its loss, as before round 3)
_T_REFRESH_WGS = [0]      # workgroups of the t_refresh launch (0: one per 64 x 64 tile, ~66 000; 256 / 96 walking the tiles measured 0.1-0.3 % slower)
_DET_PRIORITY = [0]       # default stream priority of the detector stream (the main stream: -1, higher)
_SINGLE_STREAM = [True]   # capture every phase graph without fusion_ops.fork (tools/ab_bench.py flips it)


class PhasedTrainStep(object):
    def __init__(self, model, batch, det_loss, fusion_loss, optimizer=None, use_graphs=True, det_priority=None,
                 grad_hook=None, next_batch=None, prefetch_geometry=None, eager_phases=(), reducers=None,
                 main_priority=-1, image_bwd_splits=1, buffer_broadcaster=None, coverage_every=50, text_prologue=True):
        """model: ScanQAHotPath (use_blip=True, train mode); batch: static device tensors (replayed in place);
        det_loss(data_dict) -> scalar over detector outputs; fusion_loss(data_dict) -> scalar over blip_loss /
        fused_feat; optimizer: stepped at the end of the step (None: the caller steps);
        grad_hook: optional callable run on the main stream after both backward phases and before the optimizer
        (data-parallel gradient exchange); next_batch: where the loader puts the FOLLOWING step's inputs (only its
        point_clouds are read, by the geometry prefetch).  prefetch_geometry: compute the sampling / grouping indices
        of the next batch under this step's fusion phase -- ONLY valid when next_batch really holds the next step's
        point clouds by the time this step's detector forward has run, so it defaults to on iff next_batch is given
        (a static benchmark batch passes next_batch=batch); without it the indices are computed inside the detector
        forward of the batch they belong to;
        eager_phases: names of phases launched kernel by kernel even when the rest replays from graphs (bench.py keeps
        "geometry" eager so that HIP events can bracket the FPS launch inside the timed steps; ~60 launches);
        reducers: data parallel -- {"fusion" | "image" | "det": ddp.PackedGradReducer over the parameters whose
        gradients that phase produces}: each group is exchanged on a communication stream as soon as its backward
        phase has finished (the fusion group, 3/4 of the bytes, travels under the image and detector backward) and
        the optimizer waits for all of them;
        main_priority: stream priority of the critical path (image forward -> fusion -> image backward -> optimizer);
        image_bwd_splits: data parallel -- the image encoder's backward as this many block-range phases (autograd cut in
        front of blocks depth * k / splits, vit.VisionTransformer.grad_cuts), each with its own weight-gradient flush and
        its own reducer group "image_0" (last blocks, first to finish) ... "image_{splits-1}": only the last group's
        exchange is exposed, the others travel under the rest of the image backward -- what DDP's buckets firing during
        backward do (scripts/train.py:346-347).  1 (default, single GPU): one phase, one grouped weight-gradient flush;
        buffer_broadcaster: ddp.BufferBroadcaster -- rank 0's buffers (BatchNorm running statistics) to every rank at the
        start of every step, DDP's broadcast_buffers=True (the reference's default); None: statistics stay per rank;
        coverage_every: run ddp.check_coverage every this many replayed steps (and after every capture);
        text_prologue: the token-only head of the fusion (question / answer embeddings, BLIP_VQA3D.prepare_text) and its
        backward as phases of their own on the detector stream, beside the image encoder / the image backward, instead of on
        the critical chain (~50 launches of 5 us)."""
        self.model, self.batch, self.det_loss, self.fusion_loss = model, batch, det_loss, fusion_loss
        self.opt, self.grad_hook = optimizer, grad_hook
        if prefetch_geometry and next_batch is None:
            raise ValueError("prefetch_geometry needs next_batch: the buffers the loader fills with the FOLLOWING step's "
                             "point clouds (pass next_batch=batch only for a static batch)")
        self.next_batch = next_batch if next_batch is not None else batch
        self.prefetch = (next_batch is not None) if prefetch_geometry is None else bool(prefetch_geometry)
        self.eager_phases = tuple(eager_phases)
        # weight / bias gradients of every linear are parked during a backward phase and produced by ONE grouped GEMM
        # launch per tile class + one grouped column-sum launch at its end (fusion_ops.begin/flush_deferred_wgrad)
        self.defer_wgrad = True
        self.image_splits = max(1, int(image_bwd_splits))
        vit = model.blip_model.visual_encoder
        depth = len(vit.blocks)
        # (kept HERE, not on the module: the cuts are switched on around this step's image forward only, so a plain
        # model(data_dict) + loss.backward() outside PhasedTrainStep still differentiates down to patch_embed)
        self._vit_cuts = tuple(sorted({depth * k // self.image_splits for k in range(1, self.image_splits)} - {0}))
        self.image_splits = len(self._vit_cuts) + 1
        if self.image_splits > 4:
            raise ValueError("image_bwd_splits: at most 4 block ranges")
        bm = getattr(model, "blip_model", None)
        self.text_prologue = bool(text_prologue) and bm is not None and hasattr(bm, "prepare_text") \
            and isinstance(batch.get("question"), dict)
        self._seg_probe = None   # attach_reducers: {segment: parameters whose gradient that segment produced}
        self.buffer_broadcaster, self.coverage_every, self._steps = buffer_broadcaster, int(coverage_every), 0
        self.reducers = dict(reducers or {})
        self.s_comm = torch.cuda.Stream(device=batch["point_clouds"].device) if self.reducers else None
        self.e_img_bwd = torch.cuda.Event()
        self.e_img_seg = [torch.cuda.Event() for _ in range(self.image_splits)]
        self._comm_events = []
        self.comm_stall = None   # set to [] to time how long the main stream waits for the gradient exchanges (exposed time)
        self._geo_next, self._geo_cur = None, None
        self.host_times = None  # set to {} to record the host time of every graph launch (ms, per phase)
        self.phase_events = None  # set to {} to bracket every phase with events on its stream (phase_gpu_ms())
        dev = batch["point_clouds"].device
        self.dev = dev
        # the critical path (image forward -> fusion -> image backward -> optimizer) on a HIGH-priority stream: its
        # kernels win the dispatch whenever both streams have work (measured, c3: 46.5 -> 45.5 ms; the detector stream at
        # high priority instead: 47.3 ms; this stack offers two levels, 0 and -1)
        # (the process-wide pair: fusion_state.phase_streams -- a second runner must not open more streams)
        self.s_main, self.s_det = ops.phase_streams(dev, int(main_priority),
                                                    int(_DET_PRIORITY[0] if det_priority is None else det_priority))
        self.s_img = self.s_main
        self.e_img_fwd = torch.cuda.Event()
        self.e_text_bwd, self.e_t_refresh = torch.cuda.Event(), torch.cuda.Event()
        self.t_refresh = bool(ops.TRANSPOSED_DX[0]) and bm is not None
        self.e_det_fwd, self.e_fused, self.e_det_bwd, self.e_done = (torch.cuda.Event() for _ in range(4))
        self._bn_modules, self._bn_sig = None, None
        self.use_graphs = use_graphs
        self.graphs = None
        self.loss = None
        self._state = {}

    # ---- phases (each runs entirely on one stream) -----------------------------------------------------------
    def _image_fwd(self):
        ops.new_step(self.dev)
        vit = self.model.blip_model.visual_encoder
        with vit.autograd_cuts(self._vit_cuts):
            self._state["img"] = self.model.encode_image(self.batch)
        self._state["img_cuts"], vit.cut_pairs = list(vit.cut_pairs), []   # (the module does not keep them alive)
        if len(self._state["img_cuts"]) != self.image_splits - 1:
            raise RuntimeError("image_bwd_splits=%d needs the image encoder's fused block path (no grad checkpointing, "
                               "no register_blk / return_fm): the forward made %d of %d autograd cuts"
                               % (self.image_splits, len(self._state["img_cuts"]), self.image_splits - 1))

    def _geometry(self):
        """sampling / grouping indices of the next batch, into the persistent `next` buffers"""
        from .pointnet2_utils import background_geometry
        with background_geometry():   # (the gentle ball-query grid: this phase runs beside the fusion chain)
            geo = self.model.detection_backbone.precompute_geometry(self.next_batch["point_clouds"])
        if self._geo_next is None:
            self._geo_next = {k: v.clone() for k, v in geo.items()}
        else:
            for k, v in geo.items():
                self._geo_next[k].copy_(v)

    def _det_fwd(self):
        dd = dict(self.batch)
        if self.prefetch:
            # next -> cur: the backward of this step keeps reading `cur` while the prefetch refills `next`
            if self._geo_cur is None:
                self._geo_cur = {k: v.clone() for k, v in self._geo_next.items()}
            else:
                for k, v in self._geo_next.items():
                    self._geo_cur[k].copy_(v)
            dd["geometry"] = self._geo_cur
        dd = self.model.detect_objects(dd)
        self._state["dd"] = dd

    def _text_prep(self):
        """the token-only head of the fusion (question / answer embeddings, targets): detector stream, beside the image
        encoder -- BLIP_VQA3D.prepare_text"""
        bm = self.model.blip_model
        self._state["prep"] = bm.prepare_text(self.batch["question"], self.batch.get("answer"), self.dev)

    def _t_refresh(self):
        """the K-contiguous copies of the text side's weights that this step's small input-gradient GEMMs read
        (fusion_state.transposed_shadow): ONE launch re-transposes all of them from the operands the previous step's
        optimizer left.  Detector stream, after the detection loss: it runs beside the fusion's FORWARD chain, the
        fusion's backward waits for it (at the head of the step it cost the image encoder 0.3 ms; left to the first use
        it would sit on the critical chain)."""
        if not (torch.cuda.is_current_stream_capturing() and ops._T_STATE["dirty"]):
            ops.refresh_transposed(self.dev, max_wgs=_T_REFRESH_WGS[0])

    def _text_prep_bwd(self):
        """its backward (LayerNorm + embedding tables; the decoder's table is tied to the LM head: this adds to the gradient
        the fusion phase left there): detector stream, after the fusion, beside the image backward"""
        st = self._state
        roots, seeds = [], []
        for k in ("q_embeds", "a_embeds"):
            leaf = st["prep_leaves"].get(k)
            if leaf is not None and leaf.grad is not None:
                roots.append(st["prep"][k])
                seeds.append(leaf.grad)
        if roots:
            torch.autograd.backward(roots, seeds)

    def _det_loss(self):
        """the detection loss (~200 short launches, 1 ms) as a phase of its own: the fusion waits for the detector's OUTPUTS
        (e_det_fwd), not for its loss, which then runs on the detector stream beside the start of the fusion"""
        self._state["det_loss"] = self.det_loss(self._state["dd"])

    def _fusion(self):
        st = self._state
        img_leaf = st["img"].detach().requires_grad_(True)
        obj_leaf = st["dd"]["object_feat"].detach().requires_grad_(True)
        dd = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in st["dd"].items()}
        prep = None
        if self.text_prologue:
            # the embeddings computed by the text_prep phase enter as leaves; text_prep_bwd continues from their gradients
            prep = dict(st["prep"])
            st["prep_leaves"] = {k: prep[k].detach().requires_grad_(True) for k in ("q_embeds", "a_embeds") if k in prep}
            prep.update(st["prep_leaves"])
        dd = self.model.fuse(dd, img_leaf, obj_leaf, text_prep=prep)
        st["fusion_loss_t"] = self.fusion_loss(dd)
        st["fusion_leaves"] = (img_leaf, obj_leaf)

    def _fusion_bwd(self):
        """the fusion's backward, a phase (graph) of its own behind the forward on the same stream: a captured graph can
        only wait for another stream at its head, and this one waits for t_refresh"""
        st = self._state
        loss = st.pop("fusion_loss_t")
        img_leaf, obj_leaf = st.pop("fusion_leaves")
        if self.defer_wgrad:
            ops.begin_deferred_wgrad()  # dW / db of the linears: parked, then one grouped launch after the chain
        try:
            loss.backward()
        finally:
            ops.flush_deferred_wgrad()
        st["img_grad"], st["obj_grad"] = img_leaf.grad, obj_leaf.grad
        st["fusion_loss"] = loss.detach()

    def _image_bwd(self, seg=0):
        """block range `seg` of the image encoder's backward (0 = the last blocks; one range without splits)"""
        st = self._state
        cuts = st["img_cuts"]
        if self.defer_wgrad:
            ops.begin_deferred_wgrad()  # 48 ViT weight gradients -> one grouped launch of 1296 full-contraction tiles
        try:
            if seg == 0:
                st["img"].backward(st["img_grad"])
            else:
                (x, n), (xl, nl) = cuts[len(cuts) - seg]
                torch.autograd.backward([x, n], [xl.grad, nl.grad])
        finally:
            ops.flush_deferred_wgrad()
        if seg == self.image_splits - 1 and not torch.cuda.is_current_stream_capturing():
            st["img_cuts"] = []   # (eager steps: the cut activations die with the step; captured ones live in the pool)
        if self._seg_probe is not None:   # (dry eager step of attach_reducers)
            seen = {id(p) for ps in self._seg_probe.values() for p in ps}
            self._seg_probe[seg] = [p for p in self.model.blip_model.visual_encoder.parameters()
                                    if p.grad is not None and id(p) not in seen]

    def _image_bwd_1(self):
        self._image_bwd(1)

    def _image_bwd_2(self):
        self._image_bwd(2)

    def _image_bwd_3(self):
        self._image_bwd(3)

    def _det_bwd(self):
        st = self._state
        torch.autograd.backward([st["det_loss"], st["dd"]["object_feat"]], [None, st["obj_grad"]])

    def _finish(self):
        if self.reducers and not torch.cuda.is_current_stream_capturing():
            from .ddp import check_coverage
            check_coverage(self.model, self.reducers.values())  # (eager steps only: a replayed graph runs no Python)
        if self.grad_hook is not None:
            self.grad_hook()
        if self.opt is not None:
            # (fusion_ops' optimizer post-step hook refreshes the bf16 weight shadows here)
            self.opt.step()
        st = self._state
        self.loss = st["det_loss"].detach() + st["fusion_loss"]

    # (phase, stream, memory pool): the image phases may sit on their own (CU-masked) stream but still alternate
    # strictly with the main stream's phases, so they share its pool
    _ORDER = (("text_prep", "det", "det"), ("det_fwd", "det", "det"), ("det_loss", "det", "det"), ("t_refresh", "det", "det"),
              ("geometry", "det", "det"), ("image_fwd", "img", "main"),
              ("fusion", "main", "main"), ("fusion_bwd", "main", "main"), ("text_prep_bwd", "det", "det"), ("det_bwd", "det", "det"), ("image_bwd", "img", "main"), ("image_bwd_1", "img", "main"),
              ("image_bwd_2", "img", "main"), ("image_bwd_3", "img", "main"), ("finish", "main", "main"))

    def _stream(self, which):
        return {"main": self.s_main, "det": self.s_det, "img": self.s_img}[which]

    def _skipped(self, name):
        if name.startswith("image_bwd_"):
            return int(name.rsplit("_", 1)[1]) >= self.image_splits
        if name in ("text_prep", "text_prep_bwd"):
            return not self.text_prologue
        if name == "t_refresh":
            return not self.t_refresh
        return name == "geometry" and not self.prefetch

    def phase_gpu_ms(self):
        """after a synchronize: {phase: (mean start, mean end)} in ms relative to the start of the step's first phase,
        from the events recorded while phase_events was a dict"""
        out = {}
        steps = min(len(v) for v in self.phase_events.values())
        for name, evs in self.phase_events.items():
            a = [self.phase_events["det_fwd"][i][0].elapsed_time(evs[i][0]) for i in range(steps)]
            b = [self.phase_events["det_fwd"][i][0].elapsed_time(evs[i][1]) for i in range(steps)]
            out[name] = (sum(a) / steps, sum(b) / steps)
        return out

    def _run(self, name, eager):
        if self.phase_events is not None:
            s_ = torch.cuda.current_stream(self.dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s_)
            self._run_inner(name, eager)
            e1.record(s_)
            self.phase_events.setdefault(name, []).append((e0, e1))
        else:
            self._run_inner(name, eager)

    def _run_inner(self, name, eager):
        if eager or name in self.eager_phases:
            getattr(self, "_" + name)()
        elif self.host_times is not None:
            import time
            t0 = time.perf_counter()
            self.graphs[name].replay()
            self.host_times.setdefault(name, []).append((time.perf_counter() - t0) * 1e3)
        else:
            self.graphs[name].replay()

    def _reduce(self, group, after_event):
        """exchange one gradient group on the communication stream once `after_event` (its backward) has passed"""
        r = self.reducers.get(group)
        if r is None:
            return None
        self.s_comm.wait_event(after_event)
        with torch.cuda.stream(self.s_comm):
            r.all_reduce()
            ev = torch.cuda.Event()
            ev.record(self.s_comm)
        self._comm_events.append(ev)
        return ev

    def _schedule(self, eager):
        """Launch the phases with their cross-stream dependencies (the host returns without waiting for the GPU).
        Host ORDER matters: a graph launch blocks the host while its stream's hardware queue is full, so at every
        point the detector stream's launches (short, its queue is usually empty) are issued before the main
        stream's -- otherwise the detector only gets its packets once the main stream has drained (measured)."""
        sm, sd, si = self.s_main, self.s_det, self.s_img
        # whatever the caller enqueued on ITS stream before this step -- solver.BatchStager.advance() copies the next
        # batch into the static buffers there -- happens before any phase reads the batch
        cur = torch.cuda.current_stream(self.dev)
        if self.buffer_broadcaster is not None:
            # DDP's broadcast_buffers: rank 0's buffers to every rank before the forward (after the previous step's
            # detector forward has written its BatchNorm statistics: e_done), on the caller's stream
            cur.wait_event(self.e_done)
            self.buffer_broadcaster.broadcast()
        for s_ in {sm, sd, si}:
            if s_ is not cur:
                s_.wait_stream(cur)
        sd.wait_event(self.e_done)  # parameters of the previous step's optimizer
        with torch.cuda.stream(sd):
            if self.text_prologue:
                self._run("text_prep", eager)
            self._run("det_fwd", eager)
            if _DET_LOSS_LATE[0]:
                self.e_det_fwd.record(sd)
            self._run("det_loss", eager)
            if not _DET_LOSS_LATE[0]:
                self.e_det_fwd.record(sd)
            if self.t_refresh:
                self._run("t_refresh", eager)
                self.e_t_refresh.record(sd)
            if self.prefetch:
                self._run("geometry", eager)
        if si is not sm:
            si.wait_event(self.e_done)
        with torch.cuda.stream(si):
            self._run("image_fwd", eager)
            self.e_img_fwd.record(si)
        with torch.cuda.stream(sm):
            sm.wait_event(self.e_img_fwd)
            sm.wait_event(self.e_det_fwd)
            self._run("fusion", eager)
            if self.t_refresh:
                sm.wait_event(self.e_t_refresh)
            self._run("fusion_bwd", eager)
            self.e_fused.record(sm)
        sd.wait_event(self.e_fused)
        with torch.cuda.stream(sd):
            if self.text_prologue:
                self._run("text_prep_bwd", eager)     # (the embeddings' gradients: part of the fusion group)
                self.e_text_bwd.record(sd)
            self._run("det_bwd", eager)
            self.e_det_bwd.record(sd)
        if self.text_prologue and self.s_comm is not None:
            self.s_comm.wait_event(self.e_text_bwd)
        self._reduce("fusion", self.e_fused)
        self._reduce("det", self.e_det_bwd)
        si.wait_event(self.e_fused)
        for seg in range(self.image_splits):
            with torch.cuda.stream(si):
                self._run("image_bwd" if seg == 0 else "image_bwd_%d" % seg, eager)
                self.e_img_seg[seg].record(si)
            # (one range: the group is called "image"; several: "image_0" travels under the ranges that follow it)
            self._reduce("image" if self.image_splits == 1 else "image_%d" % seg, self.e_img_seg[seg])
        with torch.cuda.stream(si):
            self.e_img_bwd.record(si)
        with torch.cuda.stream(sm):
            sm.wait_event(self.e_img_bwd)
            sm.wait_event(self.e_det_bwd)
            if self.comm_stall is not None and self._comm_events:
                # exposed communication = how long the critical path stands at this point for the exchanges alone
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(sm)
                for ev in self._comm_events:
                    sm.wait_event(ev)
                e1.record(sm)
                self.comm_stall.append((e0, e1))
            else:
                for ev in self._comm_events:
                    sm.wait_event(ev)
            del self._comm_events[:]
            self._run("finish", eager)
            self.e_done.record(sm)
        # the caller's stream is ordered behind the step (a stream-side wait, the host does not block): the returned loss /
        # the gradients can be read from it -- `pipe.eager_step().item()` used to race with the main phase stream
        cur.wait_event(self.e_done)

    def attach_reducers(self, make_reducer):
        """Data parallel: one eager step to see which parameters receive a gradient in which phase, then
        self.reducers[group] = make_reducer(list_of_parameters) for the groups "fusion", "det" and "image" (or, with
        image_bwd_splits > 1, "image_0" ... one per block range of the image backward).
        (Parameters that get no gradient on this path -- unused BLIP heads -- are left out, as DDP's
        find_unused_parameters would discover every step.)"""
        cur = torch.cuda.current_stream(self.dev)
        for s_ in (self.s_main, self.s_det, self.s_img):
            s_.wait_stream(cur)
        self.e_done.record(self.s_main)
        if self.prefetch and self._geo_next is None:
            with torch.cuda.stream(self.s_det):
                self._geometry()
        opt, self.opt = self.opt, None  # a dry step: no parameter update (replicas must not diverge)
        self._seg_probe = {}
        self.eager_step()
        probe, self._seg_probe = self._seg_probe, None
        self.opt = opt
        torch.cuda.synchronize(self.dev)
        groups = {"fusion": [], "det": []}
        if self.image_splits == 1:
            groups["image"] = probe.get(0, [])
        else:   # which block range produced which gradient was OBSERVED (a block's norm1 runs in the block before it)
            for seg in range(self.image_splits):
                groups["image_%d" % seg] = probe.get(seg, [])
        for name, p in self.model.named_parameters():
            if p.grad is None or name.startswith("blip_model.visual_encoder."):
                continue
            groups["fusion" if name.startswith("blip_model.") else "det"].append(p)
        self.reducers = {g: make_reducer(ps) for g, ps in groups.items() if ps}
        self.s_comm = torch.cuda.Stream(device=self.dev)
        return self.reducers

    def zero_grad(self):
        for p in self.model.parameters():
            p.grad = None

    def eager_step(self):
        self.zero_grad()
        self._schedule(eager=True)
        return self.loss

    def comm_report(self):
        """after a synchronize, data parallel with timing switched on (reducer.timing = [], self.comm_stall = []):
        {"groups": {name: {"bytes_on_wire", "ms"}}, "exposed_ms"} -- per-group duration of pack + collective + unpack on the
        communication stream, and the mean time the critical path stood waiting for them in front of the optimizer"""
        groups = {g: {"bytes_on_wire": r.nbytes_on_wire(), "ms": (round(r.comm_ms(), 3) if r.comm_ms() is not None else None),
                      "algo": r.algo} for g, r in self.reducers.items()}
        stall = None
        if self.comm_stall:
            stall = round(sum(a.elapsed_time(b) for a, b in self.comm_stall) / len(self.comm_stall), 3)
        return {"groups": groups, "exposed_ms": stall,
                "note": "ms = pack + collective + unpack on the communication stream (HIP events); exposed_ms = the main "
                        "stream's wait for all groups in front of the optimizer phase"}

    def _bn_momenta(self):
        """BatchNorm momentum is a HOST scalar baked into the captured launches (F.batch_norm and the bq BN kernels take
        it by value): the reference's BNMomentumScheduler (lib/pointnet2/pytorch_utils.py BNMomentumScheduler, stepped
        once per epoch) would have no effect under replay, so step() compares this signature and re-captures"""
        if self._bn_modules is None:
            self._bn_modules = [m for m in self.model.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm)]
        return tuple(m.momentum for m in self._bn_modules)

    def capture(self, warmup=3, keep_warmup_updates=False, _again=False):
        """`warmup` eager steps on the phase streams (autograd's AccumulateGrad nodes remember the stream they were
        created on -- they must be born on the stream that is later captured), then one graph per phase.  Graphs of
        one stream share a memory pool (they always replay in capture order); the two streams' pools are separate
        because their graphs run concurrently.  The warm-up steps are real optimizer steps on the first batch (the
        optimizer state must exist before the capture); unless keep_warmup_updates, parameters, buffers (BatchNorm
        running statistics), moments and the step count are put back afterwards, so training starts from the state
        the caller handed over."""
        import gc
        gc.collect()   # (dead autograd graphs of earlier eager forwards -- reference cycles -- must not lend their
        #                AccumulateGrad nodes, born on another stream, to the capture: hipStreamEndCapture faults on them)
        cur = torch.cuda.current_stream(self.dev)
        for s_ in (self.s_main, self.s_det, self.s_img):
            s_.wait_stream(cur)
        self.e_done.record(self.s_main)
        if self.prefetch and not _again:  # (a re-capture keeps the geometry the previous step prefetched)
            with torch.cuda.stream(self.s_det):
                self._geometry()  # the first step's own geometry
        snapshot, opt_snapshot = None, None
        if warmup and not keep_warmup_updates:
            torch.cuda.synchronize(self.dev)
            snapshot = {k: v.detach().clone() for k, v in self.model.state_dict().items()}
            if self.opt is not None:
                # a resumed optimizer (load_state_dict) hands over moments and a step count: they come back as they were;
                # (storage pointer -> copy: FusedAdamW's shared step counter is saved once)
                opt_snapshot = {}
                for st in self.opt.state.values():
                    for t in st.values():
                        if torch.is_tensor(t) and t.data_ptr() not in opt_snapshot:
                            opt_snapshot[t.data_ptr()] = t.detach().clone()
        for _ in range(warmup):
            self.eager_step()
        torch.cuda.synchronize(self.dev)
        if snapshot is not None:
            with torch.no_grad():
                for k, v in self.model.state_dict().items():
                    v.copy_(snapshot[k])
                if self.opt is not None:
                    done = set()
                    for st in self.opt.state.values():
                        for name, t in st.items():
                            if not torch.is_tensor(t) or t.data_ptr() in done:
                                continue
                            done.add(t.data_ptr())
                            old = opt_snapshot.get(t.data_ptr())
                            if old is not None and old.shape == t.shape:
                                t.copy_(old)
                            else:
                                t.zero_()   # state the warm-up created: a fresh optimizer starts from zero
            ops.refresh_shadows(only_with_grad=False)
            del snapshot, opt_snapshot
            torch.cuda.synchronize(self.dev)
        if not self.use_graphs:
            return self
        self.zero_grad()
        self._state = {}
        if self.t_refresh:
            ops.refresh_transposed(self.dev)   # (builds the device table of what the warm-up registered: not possible inside a capture)
            # the captured t_refresh launch reads this generation's tables: they (and the operands they point at) live as
            # long as these graphs do
            self._t_generation = ops.transposed_generation()
        pools = {"main": torch.cuda.graph_pool_handle(), "det": torch.cuda.graph_pool_handle()}
        # every phase graph SINGLE-STREAM (no fusion_ops.fork inside: the decoder's hoisted K/V projection stays on the
        # chain's stream): this runtime enqueues a graph with an internal fork node by node -- the fusion graph's launch held
        # the host for 35-38 ms and the phase ran 14.2 ms; without the fork 2.2 ms of host time and 12.7-13.4 ms (round 4,
        # A/B/A on one box: 39.5 / 40.7 / 38.6 ms per step)
        prev_overlap = ops.set_overlap(not _SINGLE_STREAM[0])
        self.graphs = {}
        self._bn_sig = self._bn_momenta()
        try:
            for name, which, pool in self._ORDER:
                if self._skipped(name) or name in self.eager_phases:
                    continue
                g = torch.cuda.CUDAGraph()
                try:
                    with torch.cuda.graph(g, pool=pools[pool], stream=self._stream(which)):
                        getattr(self, "_" + name)()
                except Exception as e:
                    raise RuntimeError("PhasedTrainStep.capture: phase '%s' could not be captured: %s" % (name, e)) from e
                self.graphs[name] = g
                torch.cuda.synchronize(self.dev)
        finally:
            ops.set_overlap(prev_overlap)
        self.e_done.record(self.s_main)
        if self.reducers:
            from .ddp import check_coverage
            check_coverage(self.model, self.reducers.values())   # the captured graphs' gradient set == the reducers' set
        return self

    def step(self):
        """one optimisation step; returns the (device) loss of this step without synchronising the host (the caller's
        current stream is ordered behind the step: reading the loss from it is safe)"""
        if self.graphs is None:
            return self.eager_step()
        if self.opt is not None and hasattr(self.opt, "sync_hyperparams"):
            self.opt.sync_hyperparams()  # LR schedulers act on param_groups; the captured step reads the pinned table
        if self._bn_momenta() != self._bn_sig:
            # a BN-momentum scheduler stepped (once per epoch in the reference): the value is baked into the graphs
            torch.cuda.synchronize(self.dev)
            self.graphs = None
            self.capture(warmup=0, _again=True)
        self._schedule(eager=False)
        if self.opt is not None:
            # the captured optimizer launch runs no Python: the post-step hook that marks the K-contiguous weight copies
            # stale did not fire, and an EAGER backward between two replays would read copies one update old
            # (tests/test_pipeline_gpu.py::test_transposed_copies_after_replayed_steps)
            ops.mark_transposed_stale()
        self._steps += 1
        if self.reducers and self.coverage_every > 0 and self._steps % self.coverage_every == 0:
            from .ddp import check_coverage
            check_coverage(self.model, self.reducers.values())   # (host-side: which .grad tensors exist)
        return self.loss

    def wait(self):
        torch.cuda.current_stream(self.dev).wait_event(self.e_done)

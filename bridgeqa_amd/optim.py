"""AdamW for the hot path: one HIP launch for all parameters (csrc/adamw.hip) that also writes the bf16 operand copies
the next forward multiplies with -- the optimizer of the reference's step (scripts/train.py:410-417,
torch.optim.AdamW) with the same arithmetic, graph-capturable (the step count lives on the device)."""
import numpy as np
import torch

from . import fusion_ops


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, grad_clip_value=None):
        """grad_clip_value: clamp every gradient element to [-v, v] inside the update kernel -- the reference's
        clip_grad_value_(parameters, 1.0) before optimizer.step() (lib/solver.py:407-409) at no extra pass; the stored
        .grad tensors are left unclamped."""
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.grad_clip_value = grad_clip_value
        betas_set = {tuple(g["betas"]) for g in self.param_groups}
        eps_set = {g["eps"] for g in self.param_groups}
        if len(betas_set) != 1 or len(eps_set) != 1:
            raise ValueError("FusedAdamW: betas and eps must be the same for every parameter group (lr / weight_decay "
                             "may differ)")
        self._step_t = None
        self.shadow_ids = set()  # parameters whose bf16 operand copy the update kernel writes (fusion_ops' refresh hook skips them)
        # one device table per SUBSET of the parameters (step(subset=...)): None = all parameters in one launch
        self._subs = {}

    def _records(self, sub):
        recs = []
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is None:
                    continue
                if sub["members"] is not None and id(p) not in sub["members"]:
                    continue
                if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous() or not p.grad.is_contiguous():
                    raise RuntimeError("FusedAdamW: parameters and gradients must be contiguous fp32 CUDA tensors")
                st = self.state[p]
                if "exp_avg" not in st:
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                # torch.optim.AdamW's state layout ({"step", "exp_avg", "exp_avg_sq"}): checkpoints round-trip with the
                # reference's optimizer_state_dict (lib/solver.py:687, scripts/train.py:449).  ONE device scalar counts
                # the updates for every parameter (the kernel reads it; a replayed HIP graph increments it).
                st["step"] = self._step(p.device)
                sh = fusion_ops.shadow_of(p)
                recs.append((p, p.grad, st["exp_avg"], st["exp_avg_sq"], sh, float(g["lr"]), float(g["weight_decay"]), g))
        return recs

    def _step(self, device):
        if self._step_t is None:
            self._step_t = torch.zeros((), dtype=torch.float32, device=device)
        return self._step_t

    def state_dict(self):
        """torch.optim.AdamW's layout with an INDEPENDENT "step" tensor per parameter: the shared device counter is an
        internal detail of the kernel -- saved as one storage, a plain torch.optim.AdamW that loads the checkpoint would
        bump it once per parameter on every step (its _foreach_add_ runs over all the "step" entries)."""
        sd = super().state_dict()
        state = {}
        for k, st in sd["state"].items():   # (the per-parameter dicts are the optimizer's own objects: copy, never mutate)
            st = dict(st)
            if torch.is_tensor(st.get("step")):
                st["step"] = st["step"].detach().clone()
            state[k] = st
        sd["state"] = state
        return sd

    def load_state_dict(self, state_dict):
        """accepts torch.optim.AdamW checkpoints: the per-parameter "step" entries collapse into the one device counter
        (their maximum: parameters that never had a gradient sit at 0 there)"""
        super().load_state_dict(state_dict)
        steps = [float(st["step"]) for st in self.state.values() if "step" in st]
        dev = next((st["exp_avg"].device for st in self.state.values() if "exp_avg" in st), None)
        if steps and dev is not None:
            self._step_t = None
            self._step(dev).fill_(max(steps))
            for st in self.state.values():
                st["step"] = self._step_t
        for sub in self._subs.values():
            sub["sig"] = None  # moments were re-allocated: rebuild the device tables

    def sync_hyperparams(self):
        """Rewrite lr / weight_decay of every table record from param_groups into the pinned staging copy.  step() does
        it itself; call this before REPLAYING a captured step (the graph holds the H2D copy node of the table and the
        kernel launch, but no Python runs): LR schedulers then take effect under graph replay."""
        for sub in self._subs.values():
            if sub["pinned"] is None:
                continue
            bufs = [sub["pinned"]] + [b for b in sub["stage"] if b is not None and b is not sub["pinned"]
                                      and b.numel() == sub["pinned"].numel()]
            lrs = np.fromiter((g["lr"] for _, g in sub["order"]), dtype=np.float32, count=len(sub["order"]))
            wds = np.fromiter((g["weight_decay"] for _, g in sub["order"]), dtype=np.float32, count=len(sub["order"]))
            for buf in bufs:   # (a captured step reads whichever staging buffer was current at capture time)
                tab = buf.numpy()[:len(sub["order"]) * 56].view(self._dtype)
                tab["lr"][:] = lrs
                tab["wd"][:] = wds

    def _build(self, sub, recs, device):
        from . import _ext
        assert _ext.ADAMW_TENSOR_BYTES == 56
        dt = self._dtype = np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("s", "<u8"), ("n", "<i8"),
                                     ("lr", "<f4"), ("wd", "<f4")])
        tab = np.zeros(len(recs), dtype=dt)
        chunks = []
        sub["order"] = [(r[0], r[7]) for r in recs]   # (parameter, group): sync_hyperparams() rewrites lr / weight_decay
        self.shadow_ids.update(id(r[0]) for r in recs if r[4] is not None)
        for i, (p, g, m, v, sh, lr, wd, _) in enumerate(recs):
            tab[i] = (p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), sh.data_ptr() if sh is not None else 0,
                      p.numel(), lr, wd)
            nchunk = (p.numel() + _ext.ADAMW_CHUNK - 1) // _ext.ADAMW_CHUNK
            chunks.append(np.stack([np.full(nchunk, i, dtype=np.int32), np.arange(nchunk, dtype=np.int32)], axis=1))
        host = np.concatenate([tab.view(np.uint8).reshape(-1), np.concatenate(chunks).reshape(-1).view(np.uint8)])
        if sub["pinned"] is None or sub["pinned"].numel() != host.size:
            # first build (eager warm-up): the staging buffers are allocated once; a rebuild under graph capture (the
            # gradients move into the graph's pool) only rewrites them and records one H2D copy node
            sub["pinned"] = torch.empty(host.size, dtype=torch.uint8).pin_memory()
            sub["devbuf"] = torch.empty(host.size, dtype=torch.uint8, device=device)
        sub["pinned"].numpy()[:] = host
        nt = len(recs) * 56
        sub["table"] = sub["devbuf"][:nt]
        sub["chunks"] = sub["devbuf"][nt:].view(torch.int32).view(-1, 2)

    @torch.no_grad()
    def step(self, closure=None, subset=None, params=None, advance=True):
        """subset=None: every parameter that has a gradient, one launch (torch.optim semantics).
        subset="name", params=<iterable>: only those parameters (the membership is fixed by the first call for that
        name; later calls may omit params) -- pipeline.PhasedTrainStep steps the fusion parameters as soon as their
        gradients are complete, under the image / detector backward, and the rest at the end of the step.  The caller
        keeps the subsets disjoint and passes advance=True to exactly ONE of a training step's subset calls -- the first
        to run: it increments the shared update counter."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        sub = self._subs.get(subset)
        if sub is None:
            if subset is not None and params is None:
                raise ValueError("FusedAdamW.step: the first call for subset %r must pass params" % (subset,))
            sub = self._subs[subset] = dict(name=subset, sig=None, table=None, chunks=None, pinned=None, devbuf=None,
                                            stage=[None, None], copied=[None, None], cur=0,
                                            order=[], members=None if subset is None else {id(p) for p in params})
        recs = self._records(sub)
        if not recs:
            return loss
        from . import _ext
        device = recs[0][0].device
        sig = tuple((r[0].data_ptr(), r[1].data_ptr(), r[2].data_ptr(), r[3].data_ptr(),
                     r[4].data_ptr() if r[4] is not None else 0) for r in recs)
        capturing = torch.cuda.is_current_stream_capturing()
        if not capturing and sub["pinned"] is not None:
            # The pinned staging copy is about to be rewritten while an earlier step's asynchronous upload may not have
            # read it yet (eager training re-allocates the gradients every step, so every step rebuilds the table; a host
            # running ahead of the GPU would hand step k the gradient pointers of step k + 1).  Two staging buffers
            # alternate: the host only waits for the upload of two steps ago.
            nxt = 1 - sub["cur"]
            if sub["copied"][nxt] is not None:
                sub["copied"][nxt].synchronize()
            if sub["stage"][nxt] is None or sub["stage"][nxt].numel() != sub["pinned"].numel():
                sub["stage"][nxt] = torch.empty(sub["pinned"].numel(), dtype=torch.uint8).pin_memory()
            sub["stage"][nxt].numpy()[:] = sub["pinned"].numpy()
            sub["pinned"], sub["cur"] = sub["stage"][nxt], nxt
        if sig != sub["sig"]:
            self._build(sub, recs, device)
            sub["sig"] = sig
        else:
            self.sync_hyperparams()
        sub["stage"][sub["cur"]] = sub["pinned"]
        # the table travels to the device on EVERY step (tens of KB): a captured step therefore always contains the
        # copy node, and lr / weight_decay written into the pinned copy reach the kernel of the next launch / replay
        sub["devbuf"].copy_(sub["pinned"], non_blocking=True)
        sub["copied"][sub["cur"]] = None
        if not capturing:
            sub["copied"][sub["cur"]] = torch.cuda.Event()
            sub["copied"][sub["cur"]].record()
        if advance:
            self._step(device).add_(1.0)
        beta1, beta2 = self.param_groups[0]["betas"]
        _ext.adamw_multi(sub["table"], sub["chunks"], self._step(device), beta1, beta2, self.param_groups[0]["eps"],
                         self.grad_clip_value)
        fusion_ops.shadows_written([r[0] for r in recs if r[4] is not None])
        return loss

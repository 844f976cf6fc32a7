"""AdamW for the hot path: one HIP launch for all parameters (csrc/adamw.hip) that also writes the bf16 operand copies
the next forward multiplies with -- the optimizer of the reference's step (scripts/train.py:410-417,
torch.optim.AdamW) with the same arithmetic, graph-capturable (the step count lives on the device)."""
import numpy as np
import torch

from . import fusion_ops


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        betas_set = {tuple(g["betas"]) for g in self.param_groups}
        eps_set = {g["eps"] for g in self.param_groups}
        if len(betas_set) != 1 or len(eps_set) != 1:
            raise ValueError("FusedAdamW: betas and eps must be the same for every parameter group (lr / weight_decay "
                             "may differ)")
        self._step_t = None
        self._sig = None
        self._table = self._chunks = None
        self._pinned = None

    def _records(self):
        recs = []
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is None:
                    continue
                if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous() or not p.grad.is_contiguous():
                    raise RuntimeError("FusedAdamW: parameters and gradients must be contiguous fp32 CUDA tensors")
                st = self.state[p]
                if not st:
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                sh = fusion_ops.shadow_of(p)
                recs.append((p, p.grad, st["exp_avg"], st["exp_avg_sq"], sh, float(g["lr"]), float(g["weight_decay"])))
        return recs

    def _build(self, recs, device):
        from . import _ext
        assert _ext.ADAMW_TENSOR_BYTES == 56
        dt = np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("s", "<u8"), ("n", "<i8"),
                       ("lr", "<f4"), ("wd", "<f4")])
        tab = np.zeros(len(recs), dtype=dt)
        chunks = []
        for i, (p, g, m, v, sh, lr, wd) in enumerate(recs):
            tab[i] = (p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), sh.data_ptr() if sh is not None else 0,
                      p.numel(), lr, wd)
            nchunk = (p.numel() + _ext.ADAMW_CHUNK - 1) // _ext.ADAMW_CHUNK
            chunks.append(np.stack([np.full(nchunk, i, dtype=np.int32), np.arange(nchunk, dtype=np.int32)], axis=1))
        host = np.concatenate([tab.view(np.uint8).reshape(-1), np.concatenate(chunks).reshape(-1).view(np.uint8)])
        if self._pinned is None or self._pinned.numel() != host.size:
            # first build (eager warm-up): the staging buffers are allocated once; a rebuild under graph capture (the
            # gradients move into the graph's pool) only rewrites them and records one H2D copy node
            self._pinned = torch.empty(host.size, dtype=torch.uint8).pin_memory()
            self._devbuf = torch.empty(host.size, dtype=torch.uint8, device=device)
        self._pinned.numpy()[:] = host
        self._devbuf.copy_(self._pinned, non_blocking=True)
        nt = len(recs) * 56
        self._table = self._devbuf[:nt]
        self._chunks = self._devbuf[nt:].view(torch.int32).view(-1, 2)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        recs = self._records()
        if not recs:
            return loss
        from . import _ext
        device = recs[0][0].device
        sig = tuple((r[0].data_ptr(), r[1].data_ptr(), r[2].data_ptr(), r[3].data_ptr(),
                     r[4].data_ptr() if r[4] is not None else 0, r[5], r[6]) for r in recs)
        if sig != self._sig:
            self._build(recs, device)
            self._sig = sig
        if self._step_t is None:
            self._step_t = torch.zeros((), dtype=torch.float32, device=device)
        self._step_t.add_(1.0)
        beta1, beta2 = self.param_groups[0]["betas"]
        _ext.adamw_multi(self._table, self._chunks, self._step_t, beta1, beta2, self.param_groups[0]["eps"])
        fusion_ops.shadows_written([r[0] for r in recs if r[4] is not None])
        return loss

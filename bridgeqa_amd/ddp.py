"""Data-parallel gradient exchange for the hot path: one process per GPU, RCCL over xGMI through
torch.distributed (backend "nccl" is RCCL on ROCm).

The reference wraps the model in DistributedDataParallel(find_unused_parameters=True) (scripts/train.py:346-347): 25 MB
buckets, a used-parameter bitmap all-reduce and a buffer broadcast every step.  Here the gradients live in a few
large FLAT buffers (parameters' .grad are views into them), so

  * the forward+backward can be replayed from one HIP graph (autograd accumulates in place into fixed addresses);
  * the exchange is a handful of large all-reduces -- xGMI is a point-to-point mesh (7 links x ~153 GB/s per GPU):
    few, large, bf16 messages, not many 25 MB fp32 ones;
  * parameters that never receive a gradient on this path (unused BLIP heads, the extra LayerNorms of
    BertOutputParallel, ...) are found once by a dry run and left out, instead of a bitmap exchange per step.
"""
import torch
import torch.distributed as dist


class FlatGradReducer(object):
    def __init__(self, params, bucket_bytes=512 << 20, comm_dtype=torch.bfloat16, process_group=None):
        """params: the parameters that DO receive gradients (see `used_parameters`), all on one device, fp32."""
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.comm_dtype = comm_dtype
        self.force = False  # run the collective even for a single rank (exercises the RCCL path on a 1-GPU box)
        self.params = list(params)
        self.buckets = []  # (flat fp32 grad buffer, comm buffer or None)
        cur, cur_n = [], 0
        limit = max(bucket_bytes // 4, 1)
        groups = []
        for p in self.params:
            if cur and cur_n + p.numel() > limit:
                groups.append(cur)
                cur, cur_n = [], 0
            cur.append(p)
            cur_n += p.numel()
        if cur:
            groups.append(cur)
        for g in groups:
            n = sum(p.numel() for p in g)
            flat = torch.zeros(n, dtype=torch.float32, device=g[0].device)
            off = 0
            for p in g:
                p.grad = flat[off:off + p.numel()].view_as(p)  # autograd now accumulates IN PLACE at a fixed address
                off += p.numel()
            comm = torch.empty(n, dtype=comm_dtype, device=flat.device) if comm_dtype != torch.float32 else None
            self.buckets.append((flat, comm))

    def zero(self):
        for flat, _ in self.buckets:
            flat.zero_()

    def all_reduce(self):
        """Average the gradients over the ranks (in place).  A no-op for a single process."""
        if self.world == 1 and not (self.force and dist.is_initialized()):
            return
        inv = 1.0 / self.world
        for flat, comm in self.buckets:
            if comm is None:
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
                flat.mul_(inv)
            else:
                torch.mul(flat, inv, out=flat)      # pre-scale in fp32, then round once to the wire format
                comm.copy_(flat)
                dist.all_reduce(comm, op=dist.ReduceOp.SUM, group=self.group)
                flat.copy_(comm)

    def nbytes_on_wire(self):
        return sum((c if c is not None else f).numel() * (c if c is not None else f).element_size()
                   for f, c in self.buckets)


class PackedGradReducer(object):
    """Gradient exchange for ONE group of parameters whose gradients become ready together (one phase of
    pipeline.PhasedTrainStep): the .grad tensors stay wherever autograd put them (so a captured backward keeps its
    "first gradient is an assignment" form -- no zero-fill, no accumulate kernel per parameter), and the exchange is

        pack  (multi-tensor fp32 -> bf16 copy into one flat wire buffer)
        all-reduce of the flat buffer (RCCL, one large message per group)
        unpack (multi-tensor bf16 -> fp32 copy back) and scale by 1 / world

    issued on whatever stream is current (PhasedTrainStep uses a communication stream, so the exchange of the
    fusion gradients runs under the image / detector backward)."""

    def __init__(self, params, comm_dtype=torch.bfloat16, process_group=None):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.force = False  # run the collective even for a single rank (exercises the RCCL path on a 1-GPU box)
        self.params = [p for p in params]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.comm = torch.empty(n, dtype=comm_dtype, device=dev)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.comm[off:off + p.numel()].view_as(p))
            off += p.numel()

    def all_reduce(self):
        if self.world == 1 and not (self.force and dist.is_initialized()):
            return
        grads = [p.grad for p in self.params]
        torch._foreach_copy_(self.views, grads)
        dist.all_reduce(self.comm, op=dist.ReduceOp.SUM, group=self.group)
        torch._foreach_copy_(grads, self.views)
        if self.world > 1:
            torch._foreach_mul_(grads, 1.0 / self.world)

    def nbytes_on_wire(self):
        return self.comm.numel() * self.comm.element_size()


def used_parameters(model, run_backward):
    """Dry run: `run_backward()` must do zero_grad(set_to_none=True) + forward + backward; returns the parameters
    that ended up with a gradient (same set on every rank: same model, same path)."""
    for p in model.parameters():
        p.grad = None
    run_backward()
    used = [p for p in model.parameters() if p.requires_grad and p.grad is not None]
    for p in model.parameters():
        p.grad = None
    return used


def broadcast_parameters(model, src=0, process_group=None):
    """What DDP's constructor does once: every replica starts from rank `src`'s parameters and buffers."""
    if not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, src, group=process_group)

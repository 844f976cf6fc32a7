"""Data-parallel gradient exchange for the hot path: one process per GPU, RCCL over xGMI through
torch.distributed (backend "nccl" is RCCL on ROCm).

The reference wraps the model in DistributedDataParallel(find_unused_parameters=True) (scripts/train.py:346-347): 25 MB
buckets, a used-parameter bitmap all-reduce and a buffer broadcast every step.  Here (PackedGradReducer)

  * the gradients stay in the tensors autograd allocated, so the backward phases replay from HIP graphs unchanged;
  * the exchange is ONE large all-reduce per backward phase -- xGMI is a point-to-point mesh (7 links x ~153 GB/s per
    GPU): few, large messages, not many 25 MB ones; fp32 on the wire by default (what the reference's DDP reduces
    in), bf16 as an explicit option (half the bytes; bench.py uses it and says so in its JSON line);
  * parameters that never receive a gradient on this path (unused BLIP heads, the extra LayerNorms of
    BertOutputParallel, ...) are found once by a dry run and left out, instead of a bitmap exchange per step;
    check_coverage() asserts that no parameter OUTSIDE the reducers ever shows up with a gradient.
"""
import torch
import torch.distributed as dist


class PackedGradReducer(object):
    """Gradient exchange for ONE group of parameters whose gradients become ready together (one phase of
    pipeline.PhasedTrainStep): the .grad tensors stay wherever autograd put them (so a captured backward keeps its
    "first gradient is an assignment" form -- no zero-fill, no accumulate kernel per parameter), and the exchange is

        pack  (multi-tensor copy of the fp32 gradients into one flat wire buffer: fp32, or bf16 when asked for)
        all-reduce of the flat buffer (RCCL, one large message per group) -- ReduceOp.AVG where the backend has it
                 (RCCL does: the division by the world size happens inside the collective), else SUM followed by ONE
                 scaling kernel over the flat buffer (gloo), never a multi-tensor pass over the fp32 gradients
        unpack (multi-tensor copy back)

    issued on whatever stream is current (PhasedTrainStep uses a communication stream, so the exchange of the
    fusion gradients runs under the image / detector backward)."""

    def __init__(self, params, comm_dtype=torch.float32, process_group=None, algo="all_reduce"):
        """algo: "all_reduce" -- one dist.all_reduce of the flat buffer (the algorithm is RCCL's choice), or
        "reduce_scatter" -- reduce-scatter + all-gather written out (on the fully connected xGMI mesh each rank
        owns 1 / world of the buffer, every link carries 2 S / world bytes and all 7 links work at once; also the form that
        lets an optimizer shard the update between the two collectives).  Same result either way."""
        if algo not in ("all_reduce", "reduce_scatter"):
            raise ValueError("PackedGradReducer: algo must be 'all_reduce' or 'reduce_scatter'")
        self.group, self.algo = process_group, algo
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.force = False  # run the collective even for a single rank (exercises the RCCL path on a 1-GPU box)
        self._avg = None    # ReduceOp.AVG available (decided at the first exchange from the group's backend)
        self.params = [p for p in params]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        # (reduce-scatter needs equal shards: the flat buffer is padded to a multiple of the world size)
        self.n = n
        npad = -(-n // self.world) * self.world if algo == "reduce_scatter" else n
        self.comm = torch.zeros(npad, dtype=comm_dtype, device=dev)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.comm[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.timing = None   # set to [] to collect (start, end) event pairs of every exchange on the stream it runs on

    def all_reduce(self):
        if self.world == 1 and not (self.force and dist.is_initialized()):
            return
        ev = None
        if self.timing is not None and self.comm.is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        grads = [p.grad for p in self.params]
        torch._foreach_copy_(self.views, grads)
        if self._avg is None:
            self._avg = dist.get_backend(self.group) == "nccl"
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        if self.algo == "reduce_scatter":
            rank = dist.get_rank(self.group)
            per = self.comm.numel() // self.world
            shard = self.comm[rank * per:(rank + 1) * per]
            dist.reduce_scatter_tensor(shard, self.comm, op=op, group=self.group)
            if not self._avg and self.world > 1:
                shard.mul_(1.0 / self.world)     # (scaling the owned shard only: 1 / world of the buffer)
            dist.all_gather_into_tensor(self.comm, shard, group=self.group)
        else:
            dist.all_reduce(self.comm, op=op, group=self.group)
            if not self._avg and self.world > 1:
                self.comm.mul_(1.0 / self.world)
        torch._foreach_copy_(grads, self.views)
        if ev is not None:
            ev[1].record()
            self.timing.append(ev)

    def comm_ms(self):
        """after a synchronize: mean duration of an exchange (pack + collective(s) + unpack) on its stream, or None"""
        if not self.timing:
            return None
        return sum(a.elapsed_time(b) for a, b in self.timing) / len(self.timing)

    def nbytes_on_wire(self):
        return self.comm.numel() * self.comm.element_size()


class BufferBroadcaster(object):
    """DDP's broadcast_buffers=True (the reference's default, scripts/train.py:346-347: before every forward rank 0's
    buffers -- the BatchNorm running statistics and num_batches_tracked of the detector -- overwrite every other rank's):
    ONE packed broadcast per dtype class per step instead of one per buffer (floating buffers travel as fp32, integer
    ones as int64).  Without it every rank keeps the running statistics of its own shard of the data; they only matter in
    eval mode, and the reference evaluates on rank 0's."""

    def __init__(self, model, src=0, process_group=None):
        self.group, self.src = process_group, src
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.force = False
        bufs = [b for b in model.buffers() if b.numel() > 0]
        self.f_bufs = [b for b in bufs if b.is_floating_point()]
        self.i_bufs = [b for b in bufs if not b.is_floating_point()]
        dev = bufs[0].device if bufs else torch.device("cpu")
        self.f_flat = torch.empty(sum(b.numel() for b in self.f_bufs), dtype=torch.float32, device=dev)
        self.i_flat = torch.empty(sum(b.numel() for b in self.i_bufs), dtype=torch.int64, device=dev)
        self.f_views, self.i_views = self._views(self.f_flat, self.f_bufs), self._views(self.i_flat, self.i_bufs)

    @staticmethod
    def _views(flat, bufs):
        out, off = [], 0
        for b in bufs:
            out.append(flat[off:off + b.numel()].view_as(b))
            off += b.numel()
        return out

    def broadcast(self):
        if self.world == 1 and not (self.force and dist.is_initialized()):
            return
        with torch.no_grad():
            for flat, views, bufs in ((self.f_flat, self.f_views, self.f_bufs), (self.i_flat, self.i_views, self.i_bufs)):
                if not bufs:
                    continue
                torch._foreach_copy_(views, bufs)
                dist.broadcast(flat, self.src, group=self.group)
                torch._foreach_copy_(bufs, views)


def check_coverage(model, reducers):
    """Raise if a parameter that no reducer exchanges holds a gradient: the reduced set was fixed by one dry run, and a
    parameter that starts receiving gradients later (a conditional path) would silently diverge between replicas."""
    covered = {id(p) for r in reducers for p in r.params}
    stray = [n for n, p in model.named_parameters() if p.grad is not None and id(p) not in covered]
    if stray:
        raise RuntimeError("data parallel: %d parameters with gradients are outside every reducer (first: %s) -- "
                           "re-run attach_reducers / used_parameters" % (len(stray), stray[0]))


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

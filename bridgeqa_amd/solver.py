"""Per-iteration hygiene of the training loop around the hot path -- SURVEY.md §8f rank 2 (reference lib/solver.py:463-595
`Solver._feed`): the places where the reference's loop stalls the GPU every iteration although no kernel needs it.

1. The running log.  `_feed` keeps 27 scalars per iteration (`_running_log`, solver.py:487-517), and for each of them does
   `.item()` (a device->host sync), `torch.tensor(value).cuda()` (a host->device copy) and, under DDP, its own
   `all_reduce` + another `.item()` (solver.py:547-556): 27 tiny collectives and ~80 synchronisations per step.
   `PackedRunningLog` stacks the 27 values on the device, all-reduces ONE vector, and brings it to the host with ONE
   copy -- the same numbers (sum over ranks / world size, fp32).

2. The batch upload.  `_feed` moves every tensor of the collated batch with a blocking `.cuda()` (solver.py:476-484) right
   before the forward.  `BatchStager` owns the static device buffers `pipeline.PhasedTrainStep` replays its graphs on
   (and the `next_batch` set its geometry prefetch reads), stages the host batch through pinned memory and copies on a
   side stream, so the upload of step n+1 runs under step n.

3. The per-iteration evaluation.  `_eval` (solver.py:437-461) runs eval_helper.get_eval -- 160 host round trips at batch 16
   -- and `.item()`s every accuracy.  bridgeqa_amd.eval_helper.get_eval(host_outputs=False) keeps everything on the device;
   `collect_running_log` gathers the 27 entries as device values for the ONE copy of item 1.

What the reference also has and this path drops on purpose: `torch.autograd.set_detect_anomaly(True)` around every
forward / backward (solver.py:524) and `CUDA_LAUNCH_BLOCKING=1` (scripts/train.py) -- debugging aids that serialise the
device; neither changes a value.
"""
import torch
import torch.distributed as dist

# solver.py:487-517, in the reference's order
RUNNING_LOG_KEYS = (
    "loss", "ref_loss", "answer_loss", "lang_loss", "objectness_loss", "vote_loss", "box_loss", "sem_cls_loss",
    "align_loss", "mae_loss",
    "ref_acc", "lang_acc", "answer_acc_at1", "answer_acc_at10", "answer_acc_at1_scene", "answer_acc_at10_scene",
    "answer_acc_at1_2d", "answer_acc_at10_2d", "answer_acc_at1_2d3d", "answer_acc_at10_2d3d",
    "answer_acc_at1_2d_over_3d", "answer_acc_at1_3d_over_2d", "obj_acc", "pos_ratio", "neg_ratio", "iou_rate_0.25",
    "iou_rate_0.5")


class PackedRunningLog(object):
    """reduce(running_log) -> {key: float}: what solver.py:547-556 appends to self.log[phase][key] for every key, from one
    stacked vector, one all-reduce (when torch.distributed is initialised) and one device->host copy."""

    def __init__(self, device, keys=RUNNING_LOG_KEYS, process_group=None):
        self.keys = tuple(keys)
        self.device = torch.device(device)
        self.group = process_group
        self._buf = torch.zeros(len(self.keys), dtype=torch.float32, device=self.device)
        self._host = torch.zeros(len(self.keys), dtype=torch.float32)
        if self.device.type == "cuda":
            self._host = self._host.pin_memory()

    def reduce(self, running_log, after=None):
        """after: the pipeline.PhasedTrainStep (anything with wait()) that produced the device values on its own streams
        -- the current stream waits for its completion event before reading them"""
        if after is not None:
            after.wait()
        vals, idx_t, idx_f, floats = [], [], [], []
        for i, k in enumerate(self.keys):
            v = running_log.get(k, 0)
            if torch.is_tensor(v):
                vals.append(v.detach().reshape(()).to(device=self.device, dtype=torch.float32))
                idx_t.append(i)
            else:
                floats.append(float(v))
                idx_f.append(i)
        if idx_f:  # python numbers: one small host->device copy for all of them
            self._buf[torch.tensor(idx_f)] = torch.tensor(floats, dtype=torch.float32).to(self.device, non_blocking=True)
        if idx_t:
            self._buf[torch.tensor(idx_t)] = torch.stack(vals)
        world = 1
        if dist.is_available() and dist.is_initialized():
            dist.all_reduce(self._buf, op=dist.ReduceOp.SUM, group=self.group)
            world = dist.get_world_size(self.group)
        self._host.copy_(self._buf, non_blocking=False)  # the one synchronisation of the iteration's logging
        out = (self._host / float(world)).tolist()
        return dict(zip(self.keys, out))


def collect_running_log(data_dict):
    """The running-log entries `_compute_loss` (solver.py:424-435) and `_eval` (:437-461) take from the data_dict of one
    iteration, WITHOUT their `.item()` / `np.mean` round trips: values stay device tensors (or python numbers where the
    dict holds them), ready for PackedRunningLog.reduce.  Works on the output of eval_helper.get_eval with
    host_outputs=False (device `ref_acc` / IoU rates) as well as with the reference's host lists."""
    def mean(v):
        if torch.is_tensor(v):
            return v.float().mean()
        return float(sum(v) / max(len(v), 1)) if isinstance(v, (list, tuple)) else float(v)

    log = {}
    for k in ("ref_loss", "answer_loss", "lang_loss", "objectness_loss", "vote_loss", "box_loss", "sem_cls_loss",
              "align_loss", "mae_loss", "loss", "lang_acc", "obj_acc", "pos_ratio", "neg_ratio"):
        if k in data_dict:
            log[k] = data_dict[k]
    if "ref_acc" in data_dict:
        log["ref_acc"] = mean(data_dict["ref_acc"])
    for k, v in data_dict.items():
        if "answer_acc" in k:
            log[k] = v
    for src, dst in (("ref_iou_rate_0.25", "iou_rate_0.25"), ("ref_iou_rate_0.5", "iou_rate_0.5")):
        if src in data_dict:
            log[dst] = mean(data_dict[src])
    return log


def _walk(d, prefix=()):
    for k, v in d.items():
        if isinstance(v, dict):
            yield from _walk(v, prefix + (k,))
        else:
            yield prefix + (k,), v


class BatchStager(object):
    """Two static device buffer sets for the tensors of a collated batch (nested dicts of tensors allowed, as the
    tokenised question / answer are; strings and lists pass through untouched, as solver.py:480-482 keeps them):
    `batch` -- what the captured graphs of pipeline.PhasedTrainStep read -- and `next_batch` -- where the FOLLOWING
    step's data is uploaded while the current step runs (and what PhasedTrainStep's geometry prefetch reads).

        stager = BatchStager(host_batch_0, device)                       # allocates both sets, uploads batch 0 into `next`
        step = PhasedTrainStep(model, stager.batch, ..., next_batch=stager.next_batch)
        for host_next in loader:                                         # host_next = data of step n + 1
            stager.advance(step)                                         # batch <- next_batch (device copy, ~0.1 ms) on
                                                                         #   the current stream, AFTER step n - 1 has finished
                                                                         #   reading `batch` (step.wait())
            stager.stage(host_next)                                      # pinned staging + H2D on a side stream: runs
            stager.wait(step.s_det)                                      #   under step n; its geometry phase waits for it
            step.step()                                                  # every phase stream first waits for the current
                                                                         #   stream, i.e. for the copy of advance()

    Stream ordering: advance() writes `batch` on the CURRENT stream; PhasedTrainStep.step() makes its phase streams wait
    for the current stream before the first phase, and advance(step) makes the current stream wait for the previous
    step's last phase (its backward graphs still read `batch`) before overwriting it.
    Shapes must not change between batches (graphs are shape-static; the reference's collate pads to fixed lengths)."""

    def __init__(self, host_batch, device):
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.stream = torch.cuda.Stream(device=self.device) if self.cuda else None
        self.e_uploaded = torch.cuda.Event() if self.cuda else None   # next_batch holds the staged data
        self.e_consumed = torch.cuda.Event() if self.cuda else None   # advance() has read next_batch
        self.batch, self.next_batch, self._pinned = {}, {}, {}
        for path, v in _walk(host_batch):
            for root in (self.batch, self.next_batch):
                node = root
                for k in path[:-1]:
                    node = node.setdefault(k, {})
                node[path[-1]] = torch.empty(v.shape, dtype=v.dtype, device=self.device) if torch.is_tensor(v) else v
            if torch.is_tensor(v):
                self._pinned[path] = torch.empty(v.shape, dtype=v.dtype).pin_memory() if self.cuda else None
        self.stage(host_batch)

    @staticmethod
    def _leaf(root, path):
        node = root
        for k in path[:-1]:
            node = node[k]
        return node, path[-1]

    def stage(self, host_batch):
        """start the upload of `host_batch` into next_batch; returns immediately on CUDA"""
        if self.cuda:
            self.e_uploaded.synchronize()          # the previous upload has left the pinned staging buffers
            self.stream.wait_event(self.e_consumed)  # ... and advance() has copied next_batch out
        for path, v in _walk(host_batch):
            node, leaf = self._leaf(self.next_batch, path)
            if path not in self._pinned:
                node[leaf] = v  # strings / lists: by reference
                continue
            dst = node[leaf]
            if tuple(v.shape) != tuple(dst.shape) or v.dtype != dst.dtype:
                raise ValueError("BatchStager: %s changed from %s %s to %s %s (graphs are shape-static)"
                                 % ("/".join(map(str, path)), tuple(dst.shape), dst.dtype, tuple(v.shape), v.dtype))
            if self.cuda:
                pin = self._pinned[path]
                pin.copy_(v)  # host -> pinned (CPU memcpy); the H2D below is then truly asynchronous
                with torch.cuda.stream(self.stream):
                    dst.copy_(pin, non_blocking=True)
            else:
                dst.copy_(v)
        if self.cuda:
            self.e_uploaded.record(self.stream)

    def wait(self, stream=None):
        """make `stream` (default: the current one) wait for the staged upload into next_batch"""
        if self.cuda:
            (stream or torch.cuda.current_stream(self.device)).wait_event(self.e_uploaded)

    def advance(self, step=None):
        """batch <- next_batch on the current stream (after the upload); non-tensor entries by reference.
        step: the pipeline.PhasedTrainStep (anything with wait()) whose in-flight phases still read `batch`: the current
        stream waits for its completion event before the copy overwrites the buffers."""
        if step is not None:
            step.wait()
        self.wait()
        src, dst = [], []
        for path in self._pinned:
            (a, ka), (b, kb) = self._leaf(self.next_batch, path), self._leaf(self.batch, path)
            src.append(a[ka])
            dst.append(b[kb])
        if dst:
            torch._foreach_copy_(dst, src)
        for path, v in _walk(self.next_batch):
            if path not in self._pinned:
                node, leaf = self._leaf(self.batch, path)
                node[leaf] = v
        if self.cuda:
            self.e_consumed.record(torch.cuda.current_stream(self.device))
        return self.batch

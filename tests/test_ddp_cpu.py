"""Multi-process data parallelism of the hot path on CPU: world_size 2, gloo, 127.0.0.1.  Each rank runs the
DET-stage hot path (bridgeqa_amd layers over the CPU oracle backend -- host-logic test) on its OWN batch;
after backward the gradients must be the mean over ranks and a step must leave both replicas identical."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from bridgeqa_amd import pointnet2_utils
        from bridgeqa_amd.hotpath import ScanQAHotPath
        from oracle import pn2_oracle
        pointnet2_utils.set_backend(pn2_oracle)
        import bench
        torch.manual_seed(0)  # identical replicas
        model = ScanQAHotPath(input_feature_dim=1, use_blip=False)
        ddp = torch.nn.parallel.DistributedDataParallel(model, find_unused_parameters=True)
        opt = torch.optim.AdamW(ddp.parameters(), lr=1e-3)
        def scenes(r):  # each rank its own scenes (DistributedSampler-like), with the synthetic detection labels
            pc = bench.synth_batch(2, 2500, 1, 42 + r, "cpu")
            return dict({"point_clouds": pc}, **bench.synth_labels(pc[..., :3], 49 + r))
        loss = bench.det_loss(ddp(scenes(rank)))
        loss.backward()
        g = model.detection_backbone.sa1.mlp_module.layer0.conv.weight.grad.clone()
        # reference: the same two batches on one replica, gradients averaged by hand
        torch.manual_seed(0)
        solo = ScanQAHotPath(input_feature_dim=1, use_blip=False)
        gs = []
        for r in range(world):
            solo.zero_grad()
            bench.det_loss(solo(scenes(r))).backward()
            gs.append(solo.detection_backbone.sa1.mlp_module.layer0.conv.weight.grad.clone())
        want = sum(gs) / world
        opt.step()
        w = model.voting_net.conv3.weight.detach().clone()
        ws = [torch.zeros_like(w) for _ in range(world)]
        dist.all_gather(ws, w)
        out.put((rank, float((g - want).abs().max()), float(want.abs().max()), float((ws[0] - ws[1]).abs().max())))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_ddp_world2_gloo_grads_are_rank_means_and_replicas_stay_equal(oracle):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=500) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, err, scale, drift in res:
        assert err <= 1e-5 * max(scale, 1.0), (rank, err, scale)
        assert drift == 0.0, (rank, drift)


def _worker_packed(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from bridgeqa_amd.ddp import PackedGradReducer, broadcast_parameters
        torch.manual_seed(rank)
        net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4))
        broadcast_parameters(net)
        x = torch.randn(5, 8, generator=torch.Generator().manual_seed(100 + rank))
        # two groups exchanged separately, as the phases of pipeline.PhasedTrainStep do
        groups = [list(net[0].parameters()), list(net[2].parameters())]
        from bridgeqa_amd.ddp import check_coverage
        reds = [PackedGradReducer(g) for g in groups]          # fp32 on the wire by default (the reference's DDP dtype)
        assert reds[0].comm.dtype == torch.float32
        net.zero_grad(set_to_none=True)
        net(x).square().mean().backward()       # .grad tensors are whatever autograd allocated
        ptrs = [p.grad.data_ptr() for p in net.parameters()]
        check_coverage(net, reds)
        try:
            check_coverage(net, reds[:1])                      # the second group's gradients are outside: must raise
            raise AssertionError("check_coverage did not raise")
        except RuntimeError:
            pass
        local = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
        for r in reversed(reds):                # backward order: last layer's group first
            r.all_reduce()
        gathered = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        want = sum(gathered) / world
        got = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
        same_storage = ptrs == [p.grad.data_ptr() for p in net.parameters()]
        bf = PackedGradReducer(groups[0], comm_dtype=torch.bfloat16)  # wire format used on the GPUs
        out.put((rank, float((got - want).abs().max()), same_storage, bf.nbytes_on_wire() == 2 * (8 * 16 + 16)))
    finally:
        dist.destroy_process_group()


def _worker_rs_and_wire(rank, world, port, out):
    """the reduce-scatter + all-gather form of the exchange equals the all-reduce form (odd sizes: the flat buffer is padded
    to whole shards), and a bf16 wire stays within 4e-3 rel-L2 of the fp32 wire's rank mean"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from bridgeqa_amd.ddp import PackedGradReducer
        g = torch.Generator().manual_seed(7 + rank)
        shapes = [(33, 17), (129,), (5, 3, 7), (1,)]                      # 561 + 129 + 105 + 1 = 796 (+ an odd total below)
        shapes.append((3,))
        res = {}
        for algo in ("all_reduce", "reduce_scatter"):
            for dt in (torch.float32, torch.bfloat16):
                ps = [torch.nn.Parameter(torch.zeros(*sh)) for sh in shapes]
                gg = torch.Generator().manual_seed(7 + rank)
                for p in ps:
                    p.grad = torch.randn(*p.shape, generator=gg) * (10.0 ** float(torch.randint(-3, 2, (1,), generator=gg)))
                r = PackedGradReducer(ps, comm_dtype=dt, algo=algo)
                ptrs = [p.grad.data_ptr() for p in ps]
                r.all_reduce()
                assert ptrs == [p.grad.data_ptr() for p in ps]
                res[(algo, dt)] = torch.cat([p.grad.reshape(-1) for p in ps])
        ref = res[("all_reduce", torch.float32)]
        rel = lambda a: float((a - ref).norm() / ref.norm())
        out.put((rank, rel(res[("reduce_scatter", torch.float32)]), rel(res[("all_reduce", torch.bfloat16)]),
                 rel(res[("reduce_scatter", torch.bfloat16)]), float(ref.abs().sum())))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_reduce_scatter_form_and_bf16_wire_world2_gloo():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_rs_and_wire, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=200) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert abs(res[0][4] - res[1][4]) < 1e-6 * res[0][4]                  # both ranks hold the same mean
    for rank, rs32, ar16, rs16, _ in res:
        assert rs32 < 1e-6, rs32                                          # same arithmetic, different collective
        assert ar16 < 4e-3 and rs16 < 4e-3, (ar16, rs16)                  # what a bf16 wire costs (two roundings)


@pytest.mark.timeout(300)
def test_packed_grad_reducer_world2_gloo():
    """per-phase exchange: gradients stay in autograd's own tensors, pack -> all-reduce -> unpack gives the rank mean"""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_packed, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=200) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, err, same_storage, wire_ok in res:
        assert err < 1e-6 and same_storage and wire_ok


def _worker_groups_and_buffers(rank, world, port, out):
    """reducer GROUPS exchanged one after the other (the image encoder's block ranges of pipeline.PhasedTrainStep) and
    ddp.BufferBroadcaster (DDP's broadcast_buffers): rank 0's BatchNorm statistics reach every rank, packed"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group(backend="gloo", init_method="env://", rank=rank, world_size=world)
    try:
        from bridgeqa_amd.ddp import BufferBroadcaster, PackedGradReducer, check_coverage
        from bridgeqa_amd.vit import VisionTransformer
        torch.manual_seed(0)
        m = VisionTransformer(img_size=32, patch_size=16, embed_dim=64, depth=4, num_heads=1, drop_path_rate=0.0).train()
        bn = torch.nn.BatchNorm1d(8)                       # buffers: running_mean / running_var / num_batches_tracked
        bn.running_mean.fill_(float(rank + 1)); bn.num_batches_tracked.fill_(10 * (rank + 1))
        holder = torch.nn.ModuleList([m, bn])
        torch.manual_seed(100 + rank)                      # every rank its own shard
        x = torch.randn(2, 3, 32, 32)
        with m.autograd_cuts((2,)):
            y = m(x)
        y.sum().backward()                                 # range 0: blocks 2-3
        seg0 = [p for p in m.parameters() if p.grad is not None]
        (xo, no), (xl, nl) = m.cut_pairs[0]
        r0 = PackedGradReducer(seg0)
        local0 = torch.cat([p.grad.reshape(-1) for p in seg0]).clone()
        r0.all_reduce()                                    # ... exchanged while range 1 runs
        torch.autograd.backward([xo, no], [xl.grad, nl.grad])
        ids0 = {id(p) for p in seg0}
        seg1 = [p for p in m.parameters() if p.grad is not None and id(p) not in ids0]
        r1 = PackedGradReducer(seg1)
        local1 = torch.cat([p.grad.reshape(-1) for p in seg1]).clone()
        r1.all_reduce()
        check_coverage(m, [r0, r1])
        errs = []
        for local, seg in ((local0, seg0), (local1, seg1)):
            gathered = [torch.zeros_like(local) for _ in range(world)]
            dist.all_gather(gathered, local)
            errs.append(float((torch.cat([p.grad.reshape(-1) for p in seg]) - sum(gathered) / world).abs().max()))
        bb = BufferBroadcaster(holder)
        bb.broadcast()
        out.put((rank, max(errs), float(bn.running_mean[0]), int(bn.num_batches_tracked), len(seg0), len(seg1)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_block_range_groups_and_buffer_broadcast_world2_gloo():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_groups_and_buffers, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=200) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, err, rm, nbt, n0, n1 in res:
        assert err < 1e-6 and n0 > 0 and n1 > 0
        assert rm == 1.0 and nbt == 10          # rank 0's statistics everywhere

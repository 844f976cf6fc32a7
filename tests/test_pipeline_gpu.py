"""bridgeqa_amd/pipeline.py: the phased two-stream schedule computes the same step as one forward + one backward.
fp32 compute, dropout off (eval-mode dropout via p=0 is not available for BatchNorm, so the model stays in train()
and the stochastic pieces are disabled by construction: hidden/attention dropout p=0, drop_path 0)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _teardown_process_group():
    """RCCL's communicator goes LAST: one full-suite run in seven aborted inside destroy_process_group() while the test's
    captured graphs (which hold collectives on the communication stream) were still alive in the test's frame"""
    import gc
    import torch.distributed as dist
    torch.cuda.synchronize()
    gc.collect()
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.destroy_process_group()


def _small_model(dev):
    from bridgeqa_amd.hotpath import ScanQAHotPath
    torch.manual_seed(0)
    m = ScanQAHotPath(input_feature_dim=4, use_blip=True, blip_kwargs=dict(image_size=64)).to(dev)
    m.train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        if hasattr(mod, "drop_prob"):
            mod.drop_prob = 0.0
    return m


def _batch(dev, B=2, N=4096):
    import bench
    class A(object):
        points, cin, image = N, 4, 64
    return bench.make_batch(A, "c3", B, 7, dev)


def test_phased_step_gradients_equal_single_backward(dev):
    import bench
    from bridgeqa_amd.pipeline import PhasedTrainStep
    model = _small_model(dev)
    batch = _batch(dev)
    for p in model.parameters():
        p.grad = None
    loss = bench.total_loss(model(dict(batch)))
    loss.backward()
    want = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    pipe = PhasedTrainStep(model, batch, bench.det_loss, bench.fusion_loss, optimizer=None, use_graphs=False)
    pipe.capture(warmup=0)
    got_loss = pipe.eager_step()
    torch.cuda.synchronize()
    assert abs(got_loss.item() - loss.item()) <= 1e-4 * abs(loss.item())
    got = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    assert set(got) == set(want)
    worst = max(((got[n] - want[n]).norm() / (want[n].norm() + 1e-12)).item() for n in want)
    assert worst < 2e-3, worst  # fp32 atomics in the detector backward (three_interpolate / group grads)


@pytest.mark.parametrize("optimizer", ["torch", "bq"])
def test_phased_step_graph_replay_trains(dev, optimizer):
    """six captured graphs on two streams: replays are finite, the loss moves, parameters change every step"""
    import bench
    from bridgeqa_amd import fusion_ops as ops
    from bridgeqa_amd.pipeline import PhasedTrainStep
    prev = ops.set_compute_dtype(torch.bfloat16)
    try:
        model = _small_model(dev)
        batch = _batch(dev)
        if optimizer == "torch":
            opt = torch.optim.AdamW(model.parameters(), lr=1e-3, fused=True, capturable=True)
        else:
            from bridgeqa_amd.optim import FusedAdamW
            opt = FusedAdamW(model.parameters(), lr=1e-3)
        pipe = PhasedTrainStep(model, batch, bench.det_loss, bench.fusion_loss, opt, use_graphs=True).capture(warmup=3)
        assert set(pipe.graphs) == {"text_prep", "det_fwd", "det_loss", "t_refresh", "image_fwd", "fusion", "fusion_bwd", "text_prep_bwd", "det_bwd",
                                    "image_bwd", "finish"}
        w = model.blip_model.visual_encoder.blocks[0].attn.qkv.weight
        w0 = w.detach().clone()
        losses = []
        for _ in range(16):
            l = pipe.step()
            pipe.wait()
            torch.cuda.synchronize()
            losses.append(l.item())
        assert all(x == x and abs(x) < 1e6 for x in losses), losses
        assert len(set(losses)) > 1 and not torch.equal(w0, w.detach())
        # one fixed batch, dropout off: the loss must go DOWN -- i.e. the optimizer's updates reach the bf16 operands
        # the next replay multiplies with (torch's fused AdamW does not bump version counters; see fusion_ops)
        assert min(losses[-3:]) < 0.9 * losses[0], losses
        # ... and the K-contiguous copies the text side's input-gradient GEMMs read (fusion_state.transposed_shadow): the
        # replayed text_prep phase re-transposes every one of them from the operands as they are at the START of the step
        regs = [e for e in ops._TSHADOW.values() if all(r() is not None for r in e[0]) and e[1].device == dev]
        assert len(regs) >= 12, len(regs)
        before = [e[1].clone() for e in regs]
        pipe.step()
        pipe.wait()
        torch.cuda.synchronize()
        assert all(torch.equal(e[2], b.t()) for e, b in zip(regs, before))
        assert any(not torch.equal(e[1], b) for e, b in zip(regs, before))   # (the step then moved the operands on)
    finally:
        ops.set_compute_dtype(prev)


def test_precomputed_geometry_is_value_neutral(dev):
    """Pointnet2Backbone.precompute_geometry + data_dict["geometry"] == the backbone computing its own FPS / ball
    query / three-NN: bit-identical features and indices (the indices are a pure function of the coordinates)."""
    from bridgeqa_amd.backbone_module import Pointnet2Backbone
    from conftest import scene
    torch.manual_seed(0)
    bb = Pointnet2Backbone(input_feature_dim=3).to(dev).eval()
    pc = scene(2, 5000, 3, 11).to(dev)
    with torch.no_grad():
        a = bb({"point_clouds": pc})
        geo = bb.precompute_geometry(pc)
        b = bb({"point_clouds": pc, "geometry": geo})
    for k in ("sa1_inds", "sa2_inds", "fp2_inds", "sa4_xyz", "sa1_features", "sa4_features", "fp2_features"):
        assert torch.equal(a[k], b[k]), k


def test_whole_hot_path_bf16_fused_vs_fp32_reference_composition(dev):
    """End to end: ScanQAHotPath (detector + ViT + twin fusion + decoder) with every fused bf16 kernel in play vs
    the SAME module in fp32 through the reference composition (dropout / stochastic depth off): LM loss within 3 %, vote loss within 5 %,
    FPS / ball-query indices identical, first-level detector features within the bf16 tolerance of SURVEY §8a (rel-L2 <= 2e-2)."""
    import bench
    from bridgeqa_amd import fusion_ops as ops
    model = _small_model(dev)
    batch = _batch(dev)
    with torch.no_grad():
        ref = model(dict(batch))
        ref_loss, ref_det = bench.fusion_loss(ref).item(), bench.det_loss(ref).item()
    prev = ops.set_compute_dtype(torch.bfloat16)
    try:
        with torch.no_grad():
            got = model(dict(batch))
            got_loss, got_det = bench.fusion_loss(got).item(), bench.det_loss(got).item()
    finally:
        ops.set_compute_dtype(prev)
    assert torch.equal(ref["sa1_inds"], got["sa1_inds"]) and torch.equal(ref["fp2_inds"], got["fp2_inds"])
    rel = lambda a, b: ((a.float() - b.float()).norm() / b.float().norm()).item()
    assert rel(got["sa1_features"], ref["sa1_features"]) < 2e-2
    # 14 bf16 conv+BatchNorm(train)+ReLU layers deep (4 SA levels + 2 FP levels), each within ~1e-2: errors compound
    assert rel(got["fp2_features"], ref["fp2_features"]) < 1e-1
    assert abs(got_loss - ref_loss) <= 3e-2 * abs(ref_loss), (got_loss, ref_loss)
    # detection: the vote loss is a continuous function of the votes -> tight; the other terms pick proposals (vote FPS)
    # and labels (distance thresholds) from the bf16-perturbed geometry -- discontinuous, so only a coarse bound
    assert abs(got["vote_loss"].item() - ref["vote_loss"].item()) <= 5e-2 * abs(ref["vote_loss"].item())
    assert abs(got_det - ref_det) <= 0.35 * abs(ref_det), (got_det, ref_det)


def test_geometry_prefetch_follows_changing_batches(dev):
    """The geometry prefetch computes the NEXT step's sampling / grouping indices one step early from `next_batch`.
    With batches that CHANGE every step (a loader writing the following step's point clouds into next_batch while the
    current step runs) every step's loss must equal the loss of the same model on the same batch computed without any
    prefetch; without an explicit next_batch the prefetch is off."""
    import bench
    from bridgeqa_amd.pipeline import PhasedTrainStep
    model = _small_model(dev)
    data = [_batch(dev, B=2, N=4096), None, None]
    for k in (1, 2):
        g = torch.Generator().manual_seed(100 + k)
        data[k] = dict(data[0])
        pc = data[0]["point_clouds"].clone()
        pc[..., :3] = (torch.rand(pc.shape[0], pc.shape[1], 3, generator=g) * torch.tensor([8.0, 8.0, 3.0])).to(dev)
        data[k]["point_clouds"] = pc
    want = []
    with torch.no_grad():
        for k in range(3):
            want.append(bench.total_loss(model(dict(data[k]))).item())
    assert abs(want[0] - want[1]) > 1e-3 * abs(want[0])  # the batches really differ
    assert PhasedTrainStep(model, data[0], bench.det_loss, bench.fusion_loss, None, use_graphs=False).prefetch is False
    with pytest.raises(ValueError):
        PhasedTrainStep(model, data[0], bench.det_loss, bench.fusion_loss, None, use_graphs=False, prefetch_geometry=True)
    cur = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in data[0].items()}
    nxt = {"point_clouds": data[0]["point_clouds"].clone()}
    pipe = PhasedTrainStep(model, cur, bench.det_loss, bench.fusion_loss, None, use_graphs=False, next_batch=nxt)
    assert pipe.prefetch is True
    pipe.capture(warmup=0)   # computes the geometry of the first batch (next_batch == batch 0 at this point)
    for k in range(3):
        # the loader: batch k is current, batch k+1 is already waiting in next_batch
        for name, v in data[k].items():
            if torch.is_tensor(v):
                cur[name].copy_(v)
        nxt["point_clouds"].copy_(data[(k + 1) % 3]["point_clouds"])
        got = pipe.eager_step()
        torch.cuda.synchronize()
        assert abs(got.item() - want[k]) <= 2e-4 * abs(want[k]), (k, got.item(), want[k])


def test_fused_adamw_eager_steps_with_a_lagging_gpu(dev):
    """eager training hands step() freshly allocated gradients every time, so every step rebuilds the pointer table in the
    pinned staging buffer and uploads it asynchronously; with the GPU far behind the host, step k's upload must still
    carry step k's pointers (the host waits for the previous upload before it rewrites the buffer)"""
    from bridgeqa_amd.optim import FusedAdamW
    torch.manual_seed(1)
    ps = [torch.randn(300, 70, device=dev), torch.randn(1000, device=dev)]
    grads = [[torch.randn_like(p) for p in ps] for _ in range(6)]
    big = torch.randn(4096, 4096, device=dev)

    def run(lag):
        params = [torch.nn.Parameter(p.clone()) for p in ps]
        opt = FusedAdamW(params, lr=1e-2, weight_decay=1e-2)
        for k in range(6):
            if lag:
                for _ in range(20):
                    big @ big                      # queue ~40 ms of work: the host is now well ahead of the GPU
            for p, g in zip(params, grads[k]):
                p.grad = g.clone()
            opt.step()
            if not lag:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        return params
    for x, y in zip(run(False), run(True)):
        assert torch.equal(x, y)


def test_fused_adamw_checkpoint_resume_and_scheduler_under_replay(dev):
    """(1) state_dict() carries the step count in torch.optim.AdamW's layout: a resumed FusedAdamW continues exactly
    like an uninterrupted one, and a torch AdamW checkpoint loads (reference: lib/solver.py:687, scripts/train.py:449).
    (2) lr written into param_groups after a HIP-graph capture reaches the replayed kernel (sync_hyperparams)."""
    from bridgeqa_amd.optim import FusedAdamW
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(300, 70, device=dev)), torch.nn.Parameter(torch.randn(1000, device=dev))]
    grads = [[torch.randn_like(p) for p in ps] for _ in range(6)]

    def run(opt, steps):
        for k in steps:
            for p, g in zip(opt.param_groups[0]["params"], grads[k]):
                p.grad = g.clone()
            opt.step()

    def fresh():
        return [torch.nn.Parameter(p.detach().clone()) for p in ps]
    a = fresh(); oa = FusedAdamW(a, lr=1e-2, weight_decay=1e-2); run(oa, range(6))
    b = fresh(); ob = FusedAdamW(b, lr=1e-2, weight_decay=1e-2); run(ob, range(3))
    sd = ob.state_dict()
    assert all(float(st["step"]) == 3.0 for st in sd["state"].values())
    c = [torch.nn.Parameter(p.detach().clone()) for p in b]; oc = FusedAdamW(c, lr=1e-2, weight_decay=1e-2)
    oc.load_state_dict({"state": {k: {n: (t.clone() if torch.is_tensor(t) else t) for n, t in st.items()}
                                  for k, st in sd["state"].items()}, "param_groups": sd["param_groups"]})
    run(oc, range(3, 6))
    for x, y in zip(a, c):
        assert torch.allclose(x, y, rtol=1e-6, atol=1e-7)
    t = fresh(); ot = torch.optim.AdamW(t, lr=1e-2, weight_decay=1e-2); run(ot, range(3))
    d = [torch.nn.Parameter(p.detach().clone()) for p in t]; od = FusedAdamW(d, lr=1e-2, weight_decay=1e-2)
    od.load_state_dict(__import__("copy").deepcopy(ot.state_dict()))  # (load_state_dict keeps device tensors by reference)
    run(ot, range(3, 6)); run(od, range(3, 6))
    for x, y in zip(t, d):
        assert torch.allclose(x, y, rtol=2e-5, atol=2e-6)
    # the other direction: a FusedAdamW checkpoint resumes a plain torch.optim.AdamW -- state_dict() emits one INDEPENDENT
    # step tensor per parameter (a shared storage would be bumped once per parameter by torch's _foreach_add_)
    assert len({st["step"].data_ptr() for st in sd["state"].values()}) == len(sd["state"])
    assert ob.state[b[0]]["step"] is ob.state[b[1]]["step"]   # (the optimizer's own state still shares the device counter)
    f = [torch.nn.Parameter(p.detach().clone()) for p in b]; of = torch.optim.AdamW(f, lr=1e-2, weight_decay=1e-2)
    of.load_state_dict(__import__("copy").deepcopy(sd))
    run(of, range(3, 6))
    for x, y in zip(a, f):
        assert torch.allclose(x, y, rtol=2e-5, atol=2e-6)
    # (2) graph replay reads the lr of param_groups at replay time
    e = fresh(); oe = FusedAdamW(e, lr=1e-2, weight_decay=0.0)
    for p, g in zip(e, grads[0]):
        p.grad = g.clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        oe.step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=s):
        oe.step()
    torch.cuda.synchronize()
    before = [p.detach().clone() for p in e]
    gr.replay(); torch.cuda.synchronize()
    assert all(not torch.equal(x, y) for x, y in zip(before, e))
    oe.param_groups[0]["lr"] = 0.0
    oe.sync_hyperparams()
    frozen = [p.detach().clone() for p in e]
    gr.replay(); torch.cuda.synchronize()
    assert all(torch.equal(x, y) for x, y in zip(frozen, e))


def test_capture_hands_a_resumed_optimizer_back_unchanged(dev):
    """PhasedTrainStep.capture(warmup > 0) runs real optimizer steps and then restores the model AND the optimizer state
    the caller handed over: a resumed optimizer (load_state_dict) keeps its moments and step count; state the warm-up
    created for a fresh optimizer starts from zero"""
    import bench
    from bridgeqa_amd import fusion_ops as ops
    from bridgeqa_amd.optim import FusedAdamW
    from bridgeqa_amd.pipeline import PhasedTrainStep
    prev = ops.set_compute_dtype(torch.bfloat16)
    try:
        model = _small_model(dev)
        batch = _batch(dev)
        opt = FusedAdamW(model.parameters(), lr=1e-3)
        pipe = PhasedTrainStep(model, batch, bench.det_loss, bench.fusion_loss, opt, use_graphs=False)
        pipe.capture(warmup=2)                         # fresh optimizer: the state exists now, zeroed
        torch.cuda.synchronize()
        w = model.blip_model.visual_encoder.blocks[0].attn.qkv.weight
        assert float(opt.state[w]["step"]) == 0.0 and not opt.state[w]["exp_avg"].any()
        for _ in range(3):
            pipe.eager_step()
        torch.cuda.synchronize()
        want_m = opt.state[w]["exp_avg"].clone()
        want_v = opt.state[w]["exp_avg_sq"].clone()
        want_w = w.detach().clone()
        assert float(opt.state[w]["step"]) == 3.0 and want_m.any()
        pipe2 = PhasedTrainStep(model, batch, bench.det_loss, bench.fusion_loss, opt, use_graphs=False)
        pipe2.capture(warmup=2)                        # "resumed": moments, step count and weights come back
        torch.cuda.synchronize()
        assert float(opt.state[w]["step"]) == 3.0
        assert torch.equal(opt.state[w]["exp_avg"], want_m) and torch.equal(opt.state[w]["exp_avg_sq"], want_v)
        assert torch.equal(w.detach(), want_w)
    finally:
        ops.set_compute_dtype(prev)


def test_bf16_wire_against_fp32_wire_through_rccl(dev):
    """what bench.py's bf16 wire does to a gradient group, bounded: PackedGradReducer over the real RCCL backend (world 1,
    forced collectives), fp32 wire = the gradients bit for bit, bf16 wire within 4e-3 rel-L2 of them (one rounding on the
    way in; the reference's DDP reduces in fp32) -- both exchange forms (all-reduce, reduce-scatter + all-gather)"""
    import os
    import torch.distributed as dist
    from bridgeqa_amd.ddp import PackedGradReducer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29655", RANK="0", WORLD_SIZE="1")
    dist.init_process_group(backend="nccl", init_method="env://", rank=0, world_size=1)
    try:
        def _body():   # (its locals -- steps, graphs, reducers -- are gone before the process group is)
            g = torch.Generator().manual_seed(11)
            shapes = [(768, 3072), (3072,), (2304, 768), (30524, 768), (5, 3, 7), (1,)]
            for algo in ("all_reduce", "reduce_scatter"):
                for dt, bound in ((torch.float32, 0.0), (torch.bfloat16, 4e-3)):
                    ps = [torch.nn.Parameter(torch.zeros(*sh, device=dev)) for sh in shapes]
                    want = []
                    for p in ps:
                        p.grad = (torch.randn(*p.shape, generator=g) * 10.0 ** float(torch.randint(-4, 1, (1,), generator=g))).to(dev)
                        want.append(p.grad.clone())
                    r = PackedGradReducer(ps, comm_dtype=dt, algo=algo)
                    r.force = True
                    r.timing = []
                    r.all_reduce()
                    torch.cuda.synchronize()
                    num = sum((p.grad - w).double().pow(2).sum().item() for p, w in zip(ps, want))
                    den = sum(w.double().pow(2).sum().item() for w in want)
                    assert (num / den) ** 0.5 <= bound, (algo, dt, (num / den) ** 0.5)
                    assert r.comm_ms() is not None and r.comm_ms() > 0
        _body()
    finally:
        _teardown_process_group()


def test_data_parallel_step_on_rccl_world_1_equals_the_plain_step(dev):
    """Multi-GPU readiness on a one-GPU box: the phased step with its per-phase PackedGradReducers on a communication
    stream, over the REAL RCCL backend at world size 1 (forced collectives), produces the same loss and the same
    gradients as the step without any exchange; the three groups cover every parameter that receives a gradient
    (reference: DistributedDataParallel of scripts/train.py:346-347)."""
    import os
    import torch.distributed as dist
    import bench
    from bridgeqa_amd.ddp import PackedGradReducer, check_coverage
    from bridgeqa_amd.pipeline import PhasedTrainStep
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29653", RANK="0", WORLD_SIZE="1")
    dist.init_process_group(backend="nccl", init_method="env://", rank=0, world_size=1)
    try:
        def _body():   # (its locals -- steps, graphs, reducers -- are gone before the process group is)
            model = _small_model(dev)
            batch = _batch(dev)
            plain = PhasedTrainStep(model, batch, bench.det_loss, bench.fusion_loss, optimizer=None, use_graphs=False)
            plain.capture(warmup=0)
            want_loss = plain.eager_step().item()
            torch.cuda.synchronize()
            want = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
            dp = PhasedTrainStep(model, batch, bench.det_loss, bench.fusion_loss, optimizer=None, use_graphs=False)

            def make(ps):
                r = PackedGradReducer(ps)       # fp32 on the wire
                r.force = True                  # run the collective although the world has one rank
                return r
            reds = dp.attach_reducers(make)
            assert set(reds) == {"fusion", "image", "det"}
            dp.capture(warmup=0)
            got_loss = dp.eager_step().item()
            dp.wait()
            torch.cuda.synchronize()
            check_coverage(model, reds.values())
            got = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
            assert set(got) == set(want)
            assert abs(got_loss - want_loss) <= 1e-4 * abs(want_loss)
            worst = max(((got[n] - want[n]).norm() / (want[n].norm() + 1e-12)).item() for n in want)
            assert worst < 2e-3, worst          # fp32 atomics of the detector backward; the all-reduce itself is exact at world 1
            assert sum(r.nbytes_on_wire() for r in reds.values()) == 4 * sum(p.numel() for p in model.parameters() if p.grad is not None)
        _body()
    finally:
        _teardown_process_group()


def test_split_image_backward_groups_and_buffer_broadcast_on_rccl_world_1(dev):
    """image_bwd_splits=3: the image encoder's backward as three block-range phases with their own weight-gradient flush and
    reducer group (only the last group's exchange is exposed), plus ddp.BufferBroadcaster at the start of the step -- over
    RCCL at world 1 with forced collectives: same loss and gradients as the plain step, groups disjoint and complete, and
    the same under graph replay."""
    import os
    import torch.distributed as dist
    import bench
    from bridgeqa_amd.ddp import BufferBroadcaster, PackedGradReducer, check_coverage
    from bridgeqa_amd.pipeline import PhasedTrainStep
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29654", RANK="0", WORLD_SIZE="1")
    dist.init_process_group(backend="nccl", init_method="env://", rank=0, world_size=1)
    try:
        def _body():   # (its locals -- steps, graphs, reducers -- are gone before the process group is)
            model = _small_model(dev)
            batch = _batch(dev)
            plain = PhasedTrainStep(model, batch, bench.det_loss, bench.fusion_loss, optimizer=None, use_graphs=False)
            plain.capture(warmup=0)
            want_loss = plain.eager_step().item()
            torch.cuda.synchronize()
            want = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
            bb = BufferBroadcaster(model)
            bb.force = True
            dp = PhasedTrainStep(model, batch, bench.det_loss, bench.fusion_loss, optimizer=None, use_graphs=True,
                                 image_bwd_splits=3, buffer_broadcaster=bb, coverage_every=2)
            assert dp._vit_cuts == (4, 8) and model.blip_model.visual_encoder.grad_cuts == ()   # scoped, not module state

            def make(ps):
                r = PackedGradReducer(ps)
                r.force = True
                return r
            reds = dp.attach_reducers(make)
            assert set(reds) == {"fusion", "det", "image_0", "image_1", "image_2"}
            ids = [id(p) for r in reds.values() for p in r.params]
            assert len(ids) == len(set(ids))                                    # disjoint
            names = {id(p): n for n, p in model.named_parameters()}
            g0 = {names[id(p)] for p in reds["image_0"].params}
            g2 = {names[id(p)] for p in reds["image_2"].params}
            assert any(".blocks.11." in n for n in g0) and any(".blocks.0." in n for n in g2)
            assert any("patch_embed" in n for n in g2) and not any(".blocks.0." in n for n in g0)
            got_loss = dp.eager_step().item()
            dp.wait()
            torch.cuda.synchronize()
            check_coverage(model, reds.values())
            got = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
            assert set(got) == set(want)
            assert abs(got_loss - want_loss) <= 1e-4 * abs(want_loss)
            worst = max((((got[n] - want[n]).norm() / (want[n].norm() + 1e-12)).item(), n) for n in want)
            assert worst[0] < 4e-3, worst   # (fp32 atomics in the detector's scatter gradients: 1-2e-3 between two eager executions)
            dp.capture(warmup=1)                                                # graphs: image_bwd, image_bwd_1, image_bwd_2
            assert {"image_bwd", "image_bwd_1", "image_bwd_2"} <= set(dp.graphs) and "image_bwd_3" not in dp.graphs
            for _ in range(4):                                                  # (coverage check every 2nd replayed step)
                l = dp.step()
            dp.wait()
            torch.cuda.synchronize()
            assert abs(l.item() - want_loss) <= 2e-3 * abs(want_loss), (l.item(), want_loss)
            got = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
            worst = max((((got[n] - want[n]).norm() / (want[n].norm() + 1e-12)).item(), n) for n in want)
            assert worst[0] < 4e-3, worst
            # a plain forward + backward AFTER the split step was built still reaches the patch embedding
            model.zero_grad(set_to_none=True)
            dd = model(dict(batch))
            (bench.det_loss(dd) + bench.fusion_loss(dd)).backward()
            torch.cuda.synchronize()
            assert model.blip_model.visual_encoder.patch_embed.proj.weight.grad is not None
            assert model.blip_model.visual_encoder.blocks[0].attn.qkv.weight.grad is not None
        _body()
    finally:
        _teardown_process_group()


def test_bn_momentum_change_recaptures_the_graphs(dev):
    """BatchNorm momentum is baked into the captured launches; after a BN-momentum scheduler step the phased
    step must re-capture -- with momentum 0 the running statistics must stop moving, the loss keeps going down"""
    import bench
    from bridgeqa_amd import fusion_ops as ops
    from bridgeqa_amd.optim import FusedAdamW
    from bridgeqa_amd.pipeline import PhasedTrainStep
    from bridgeqa_amd.pytorch_utils import BNMomentumScheduler
    prev = ops.set_compute_dtype(torch.bfloat16)
    try:
        model = _small_model(dev)
        batch = _batch(dev)
        opt = FusedAdamW(model.parameters(), lr=1e-3)
        pipe = PhasedTrainStep(model, batch, bench.det_loss, bench.fusion_loss, opt, use_graphs=True).capture(warmup=3)
        bn = [m for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d)][0]
        sched = BNMomentumScheduler(model, lambda it: 0.5 if it < 1 else 0.0, last_epoch=-1)  # epoch 0: 0.5, then 0.0
        graphs0 = pipe.graphs
        means, losses = [], []
        for it in range(7):
            if it == 3:
                sched.step(1)   # momentum -> 0.0
            l = pipe.step()
            pipe.wait()
            torch.cuda.synchronize()
            losses.append(l.item())
            means.append(bn.running_mean.detach().clone())
        assert pipe.graphs is not graphs0                      # re-captured
        assert not torch.equal(means[0], means[1]) and not torch.equal(means[1], means[2])   # momentum 0.5: moving
        assert torch.equal(means[3], means[2]) and torch.equal(means[5], means[2])           # momentum 0.0: frozen
        assert all(x == x for x in losses) and losses[-1] < losses[0]
    finally:
        ops.set_compute_dtype(prev)


def test_hot_path_with_qa_heads_and_full_get_loss(dev):
    """SURVEY §8f rank 1 wired end to end: ScanQAHotPath(use_lang_cls, use_reference) writes lang_scores / cluster_ref /
    decoder_loss, loss_helper.get_loss (reference / language / answer + detection terms, the reference's weights of
    scripts/train.py) consumes them, one backward reaches the new heads, the detector and the fusion; bf16 kernel path"""
    import bench
    from bridgeqa_amd import fusion_ops as ops
    from bridgeqa_amd.hotpath import ScanQAHotPath
    from bridgeqa_amd.loss_helper import get_loss
    prev = ops.set_compute_dtype(torch.bfloat16)
    try:
        torch.manual_seed(0)
        m = ScanQAHotPath(input_feature_dim=4, use_blip=True, blip_kwargs=dict(image_size=64), use_lang_cls=True,
                          use_reference=True).to(dev)
        m.train()
        keys = set(m.state_dict())
        for k in ("lang_cls.0.weight", "object_cls.3.bias", "linear_blip_to_object.weight",
                  "dec_list_qo.1.mhatt2.linear_merge.weight", "dec_list_qo.0.norm3.a_2", "enc_list_o.1.ffn.mlp.fc.linear.bias"):
            assert k in keys, k   # ScanQA's own names (qa_module.py:223-249)
        batch = _batch(dev)
        B = batch["point_clouds"].shape[0]
        dd = m(dict(batch))
        assert dd["lang_scores"].shape == (B, 18) and dd["cluster_ref"].shape == (B, 256) and "decoder_loss" in dd
        g = torch.Generator().manual_seed(1)
        dd["ref_center_label"] = dd["center_label"][:, 0]
        dd["ref_heading_class_label"] = dd["heading_class_label"][:, 0]
        dd["ref_heading_residual_label"] = dd["heading_residual_label"][:, 0]
        dd["ref_size_class_label"] = dd["size_class_label"][:, 0]
        dd["ref_size_residual_label"] = dd["size_residual_label"][:, 0]
        dd["ref_obj_mask"] = torch.ones(B, device=dev)
        dd["object_cat"] = torch.randint(0, 18, (B,), generator=g).to(dev)
        weights = dict(bench.DET_LOSS_WEIGHTS, ref_loss=0.1, lang_loss=0.1, answer_loss=1.0)
        loss, dd = get_loss(dd, bench.det_config(), detection=True, use_reference=True, use_lang_classifier=True,
                            use_answer=True, loss_weights=weights)
        assert torch.isfinite(loss) and dd["cluster_labels"].sum().item() == B      # one referred proposal per scene
        assert abs(dd["answer_loss"].item() - dd["blip_loss"].item()) < 1e-6
        loss.backward()
        for name in ("lang_cls.0.weight", "object_cls.0.weight", "dec_list_qo.0.mhatt2.linear_q.weight",
                     "linear_blip_to_object.weight", "object_feat_linear.0.weight",
                     "detection_backbone.sa1.mlp_module.layer0.conv.weight",
                     "blip_model.text_encoder.encoder.layer.0.attention.self.query.weight"):
            p = dict(m.named_parameters())[name]
            assert p.grad is not None and torch.isfinite(p.grad).all() and p.grad.abs().sum() > 0, name
        assert dict(m.named_parameters())["enc_list_o.0.mhatt.linear_q.weight"].grad is None   # never called upstream either
        # ... and the iteration's evaluation + log (lib/solver.py:437-461, :547-556) without a host round trip until the
        # ONE copy of the packed log: get_eval on the device, 27 entries through PackedRunningLog
        from bridgeqa_amd.eval_helper import get_eval
        from bridgeqa_amd.solver import RUNNING_LOG_KEYS, PackedRunningLog, collect_running_log
        dd = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in dd.items()}
        dd["ref_box_label"] = torch.nn.functional.one_hot(torch.zeros(B, dtype=torch.long, device=dev), dd["center_label"].shape[1])
        dd["answer_scores"] = torch.randn(B, 40, generator=g).to(dev)
        dd["answer_cats"] = torch.nn.functional.one_hot(torch.randint(0, 40, (B,), generator=g), 40).float().to(dev)
        get_eval(dict(dd), bench.det_config(), use_lang_classifier=True, host_outputs=False)     # warm-up (cached constants)
        torch.cuda.synchronize()
        torch.cuda.set_sync_debug_mode("error")
        try:
            ev = get_eval(dd, bench.det_config(), use_reference=True, use_lang_classifier=True, host_outputs=False)
            log = collect_running_log(ev)
        finally:
            torch.cuda.set_sync_debug_mode("default")
        vals = PackedRunningLog(dev).reduce(log)
        assert set(vals) == set(RUNNING_LOG_KEYS)
        assert abs(vals["loss"] - loss.item()) < 1e-4 * max(1.0, abs(loss.item())) and 0.0 <= vals["ref_acc"] <= 1.0
        assert 0.0 <= vals["iou_rate_0.25"] <= 1.0 and vals["iou_rate_0.5"] <= vals["iou_rate_0.25"]
        assert 0.0 <= vals["obj_acc"] <= 1.0 and 0.0 <= vals["answer_acc_at10"] <= 1.0
    finally:
        ops.set_compute_dtype(prev)


def test_batch_stager_feeds_replayed_graphs_with_changing_batches(dev):
    """solver.BatchStager + PhasedTrainStep under graph replay: host batches that change every step go
    through pinned staging into next_batch while the previous step runs, advance() hands them to the graphs' static
    buffers, the geometry prefetch reads next_batch -- the replayed losses equal plain eager forwards on the same data"""
    import bench
    from bridgeqa_amd.pipeline import PhasedTrainStep
    from bridgeqa_amd.solver import BatchStager, PackedRunningLog
    model = _small_model(dev)
    base = _batch(dev, B=2, N=4096)
    host = []
    for k in range(4):
        g = torch.Generator().manual_seed(200 + k)
        hb = {n: ({m: t.cpu() for m, t in v.items()} if isinstance(v, dict) else (v.cpu() if torch.is_tensor(v) else v))
              for n, v in base.items()}
        pc = hb["point_clouds"].clone()
        pc[..., :3] = torch.rand(pc.shape[0], pc.shape[1], 3, generator=g) * torch.tensor([8.0, 8.0, 3.0])
        hb["point_clouds"] = pc
        host.append(hb)
    want = []
    with torch.no_grad():
        for hb in host:
            dd = {n: ({m: t.to(dev) for m, t in v.items()} if isinstance(v, dict) else (v.to(dev) if torch.is_tensor(v) else v))
                  for n, v in hb.items()}
            want.append(bench.total_loss(model(dd)).item())
    assert abs(want[0] - want[1]) > 1e-3 * abs(want[0])
    st = BatchStager(host[0], dev)
    st.advance()
    pipe = PhasedTrainStep(model, st.batch, bench.det_loss, bench.fusion_loss, None, use_graphs=True,
                           next_batch=st.next_batch)
    pipe.capture(warmup=1)      # (no optimizer: the parameters stay what `want` was computed with)
    log = PackedRunningLog(dev)
    for k in range(4):
        if k > 0:
            st.advance(pipe)                          # batch <- data k (uploaded during step k - 1), after step k - 1
        st.stage(host[(k + 1) % 4])                   # data k + 1 -> next_batch, under this step
        st.wait(pipe.s_det)                           # the geometry prefetch of this step reads next_batch
        loss = pipe.step()                            # (its phase streams wait for the copy of advance() themselves)
        got = log.reduce({"loss": loss, "pos_ratio": 0.25}, after=pipe)
        assert abs(got["loss"] - want[k]) <= 5e-4 * abs(want[k]), (k, got["loss"], want[k])
        assert got["pos_ratio"] == 0.25 and got["mae_loss"] == 0.0

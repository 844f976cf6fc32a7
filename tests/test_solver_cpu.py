"""bridgeqa_amd/solver.py: the packed running log equals the reference loop's per-key reduction
(lib/solver.py:547-556) with ONE collective, and the batch stager's buffer protocol."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _reference_per_key(running_log, keys, world_values=None):
    """solver.py:547-556: value.item(), then (under DDP) one all_reduce(SUM) per key and / world"""
    out = {}
    for k in keys:
        v = running_log.get(k, 0)
        v = v.item() if torch.is_tensor(v) else v
        if world_values is not None:
            v = sum(w[k] for w in world_values) / len(world_values)
        out[k] = float(v)
    return out


def _log_of(rank):
    from bridgeqa_amd.solver import RUNNING_LOG_KEYS
    g = torch.Generator().manual_seed(100 + rank)
    log = {}
    for i, k in enumerate(RUNNING_LOG_KEYS):
        v = torch.rand((), generator=g) * (i + 1)
        log[k] = v if i % 3 else float(v)        # tensors and python floats mixed, as _running_log holds them
    del log["mae_loss"]                          # a key the iteration did not set: counts as 0 (solver.py:487)
    return log


def test_packed_running_log_single_process():
    from bridgeqa_amd.solver import RUNNING_LOG_KEYS, PackedRunningLog
    assert len(RUNNING_LOG_KEYS) == 27 and RUNNING_LOG_KEYS[0] == "loss" and RUNNING_LOG_KEYS[-1] == "iou_rate_0.5"
    log = _log_of(0)
    got = PackedRunningLog("cpu").reduce(log)
    want = _reference_per_key(log, RUNNING_LOG_KEYS)
    assert list(got) == list(RUNNING_LOG_KEYS)
    for k in RUNNING_LOG_KEYS:
        assert abs(got[k] - want[k]) <= 1e-6 * max(1.0, abs(want[k])), k


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from bridgeqa_amd.solver import PackedRunningLog
        calls = []
        real = dist.all_reduce
        dist.all_reduce = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
        got = PackedRunningLog("cpu").reduce(_log_of(rank))
        dist.all_reduce = real
        out[rank] = (got, len(calls))
    finally:
        dist.destroy_process_group()


def test_packed_running_log_world2_gloo_one_collective():
    from bridgeqa_amd.solver import RUNNING_LOG_KEYS
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = mp.Manager().dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    per_rank = [{k: (v.item() if torch.is_tensor(v) else v) for k, v in _log_of(r).items()} for r in range(2)]
    for r in range(2):
        got, ncalls = out[r]
        assert ncalls == 1                               # the reference issues 27
        for k in RUNNING_LOG_KEYS:
            want = sum(d.get(k, 0) for d in per_rank) / 2
            assert abs(got[k] - want) <= 1e-6 * max(1.0, abs(want)), (r, k)


def test_batch_stager_protocol_cpu():
    from bridgeqa_amd.solver import BatchStager
    mk = lambda seed: {"point_clouds": torch.randn(2, 50, 7, generator=torch.Generator().manual_seed(seed)),
                       "question": {"input_ids": torch.randint(0, 99, (2, 6), generator=torch.Generator().manual_seed(seed)),
                                    "attention_mask": torch.ones(2, 6, dtype=torch.long)},
                       "scene_id_str": ["scene%04d" % seed, "x"]}
    b0, b1 = mk(0), mk(1)
    st = BatchStager(b0, "cpu")
    batch = st.advance()
    assert batch is st.batch and torch.equal(batch["point_clouds"], b0["point_clouds"])
    assert torch.equal(batch["question"]["input_ids"], b0["question"]["input_ids"]) and batch["scene_id_str"] == b0["scene_id_str"]
    ptr = batch["point_clouds"].data_ptr()
    st.stage(b1)                                             # lands in next_batch; `batch` still holds step 0
    assert torch.equal(st.batch["point_clouds"], b0["point_clouds"])
    assert torch.equal(st.next_batch["point_clouds"], b1["point_clouds"])
    st.advance()
    assert st.batch["point_clouds"].data_ptr() == ptr       # static buffers: graphs keep their addresses
    assert torch.equal(st.batch["point_clouds"], b1["point_clouds"]) and st.batch["scene_id_str"] == b1["scene_id_str"]
    bad = mk(2)
    bad["point_clouds"] = torch.randn(2, 51, 7)
    with pytest.raises(ValueError):
        st.stage(bad)


def test_collect_running_log_matches_the_reference_mapping():
    """solver.py:424-461: which data_dict entries land under which running-log key"""
    from bridgeqa_amd.solver import RUNNING_LOG_KEYS, PackedRunningLog, collect_running_log
    d = {"loss": torch.tensor(3.0), "ref_loss": torch.tensor(0.5), "answer_loss": torch.tensor(1.5), "lang_loss": torch.tensor(0.25),
         "objectness_loss": torch.tensor(0.1), "vote_loss": torch.tensor(0.2), "box_loss": torch.tensor(0.3),
         "sem_cls_loss": torch.tensor(0.4), "align_loss": torch.tensor(0.0), "mae_loss": torch.tensor(0.0),
         "ref_acc": [1.0, 0.0, 0.0, 1.0], "lang_acc": torch.tensor(0.75), "answer_acc_at1": torch.tensor(0.5),
         "answer_acc_at10": torch.tensor(1.0), "obj_acc": torch.tensor(0.9), "pos_ratio": torch.tensor(0.3),
         "neg_ratio": torch.tensor(0.6), "ref_iou_rate_0.25": 0.5, "ref_iou_rate_0.5": 0.25, "ref_iou": [0.1, 0.6],
         "pred_bboxes": None}
    log = collect_running_log(d)
    assert set(log) <= set(RUNNING_LOG_KEYS)
    assert log["ref_acc"] == 0.5 and log["iou_rate_0.25"] == 0.5 and log["iou_rate_0.5"] == 0.25
    vals = PackedRunningLog("cpu").reduce(log)
    assert vals["loss"] == 3.0 and vals["answer_acc_at10"] == 1.0 and vals["answer_acc_at1_2d"] == 0.0
    # device form of get_eval: ref_acc / rates are tensors
    d["ref_acc"], d["ref_iou_rate_0.25"] = torch.tensor([1.0, 0.0, 0.0, 0.0]), torch.tensor(0.75, dtype=torch.float64)
    vals = PackedRunningLog("cpu").reduce(collect_running_log(d))
    assert vals["ref_acc"] == 0.25 and vals["iou_rate_0.25"] == 0.75

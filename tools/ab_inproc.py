"""A/B of two PhasedTrainStep configurations inside ONE process, interleaved: the pool's boxes drift by +-1 ms per step
between (and within) runs, which buries a 0.2 ms effect in back-to-back bench.py runs.  Both steps are captured once (module
switches are read at capture time), then run in alternating blocks; reported: mean / min ms per step of each, the paired
difference per block and the mean fusion-phase span.

    python tools/ab_inproc.py [--a NAME=VALUE ...] [--b NAME=VALUE ...] [--a-arg k=v ...] [--b-arg k=v ...] [--blocks 6 --steps 10]

NAME=VALUE: attribute of a bridgeqa_amd module as in tools/ab_bench.py (list switches: NAME[0]=VALUE); k=v: keyword
argument of PhasedTrainStep (e.g. fusion_bwd_cut=6)."""
import argparse
import ast
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def apply(items):
    undo = []
    for item in items:
        name, value = item.split("=", 1)
        mod, attr = name.split(".", 1)
        m = importlib.import_module("bridgeqa_amd." + mod)
        v = ast.literal_eval(value)
        if attr.endswith("[0]"):
            lst = getattr(m, attr[:-3])
            undo.append((lst, 0, lst[0], True))
            lst[0] = v
        else:
            undo.append((m, attr, getattr(m, attr), False))
            setattr(m, attr, v)
    return undo


def revert(undo):
    for obj, key, old, is_list in reversed(undo):
        if is_list:
            obj[key] = old
        else:
            setattr(obj, key, old)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--a", nargs="*", default=[])
    ap.add_argument("--b", nargs="*", default=[])
    ap.add_argument("--a-arg", nargs="*", default=[])
    ap.add_argument("--b-arg", nargs="*", default=[])
    ap.add_argument("--blocks", type=int, default=6)
    ap.add_argument("--steps", type=int, default=10)
    a = ap.parse_args()
    import bench
    from bridgeqa_amd import fusion_ops
    from bridgeqa_amd.optim import FusedAdamW
    from bridgeqa_amd.pipeline import PhasedTrainStep
    sys.argv = ["bench.py"]
    args = bench.parse()
    dev = torch.device("cuda:0")
    fusion_ops.set_compute_dtype(torch.bfloat16)
    torch.manual_seed(0)
    model = bench.build_model("c3", args.cin, args.image).to(dev)
    batch = bench.make_batch(args, "c3", args.batch, 42, dev)
    opt = FusedAdamW(model.parameters(), lr=5e-4, weight_decay=1e-5, grad_clip_value=1.0)
    pipes = []
    for sw, kw in ((a.a, a.a_arg), (a.b, a.b_arg)):
        undo = apply(sw)
        kwargs = {k: ast.literal_eval(v) for k, v in (x.split("=", 1) for x in kw)}
        p = PhasedTrainStep(model, batch, bench.det_loss, bench.fusion_loss, opt, use_graphs=True, next_batch=batch, **kwargs)
        p.capture(warmup=3)
        for _ in range(3):
            p.step()
        p.phase_events = {}
        revert(undo)
        pipes.append(p)
    torch.cuda.synchronize()
    times = [[], []]
    for blk in range(a.blocks):
        for i in (0, 1) if blk % 2 == 0 else (1, 0):
            p = pipes[i]
            p.step(); p.wait(); torch.cuda.synchronize()       # (one settling step after the switch)
            t0 = time.perf_counter()
            for _ in range(a.steps):
                p.step()
            p.wait()
            torch.cuda.synchronize()
            times[i].append((time.perf_counter() - t0) / a.steps * 1e3)
    for i, name in enumerate("AB"):
        t = times[i]
        ph = pipes[i].phase_gpu_ms()
        fus = ph["fusion_b"][1] - ph["fusion"][0] if "fusion_b" in ph else ph["fusion"][1] - ph["fusion"][0]
        print("%s: mean %.3f  min %.3f  blocks %s  fusion span %.2f ms  [%s %s]" % (
            name, sum(t) / len(t), min(t), " ".join("%.2f" % x for x in t), fus, (a.a, a.b)[i], (a.a_arg, a.b_arg)[i]))
    d = [y - x for x, y in zip(times[0], times[1])]
    print("B - A per block: %s   mean %.3f ms" % (" ".join("%+.2f" % x for x in d), sum(d) / len(d)))


if __name__ == "__main__":
    main()

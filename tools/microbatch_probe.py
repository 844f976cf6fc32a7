"""VERDICT r4 item 4 -- probe, not product: would a two-micro-batch image / fusion pipeline pay?

The c3 step's critical chain is serial: image forward 7.7 -> fusion forward + backward 11.1 -> backward window 14.6 ms, and the
fusion chain (M = 160-640 rows, ~60 workgroups per kernel) leaves most of the chip idle.  ViT and MED have no cross-sample
coupling, so the image + fusion side could run as two micro-batches of 8: fusion(mb0) beside image_fwd(mb1), image_bwd(mb0)
beside fusion(mb1).  Before building that, measure what the two graphs cost beside each other:

  two PhasedTrainStep captures at B = 8 on ONE model (separate pools), then
    a)  image_fwd(B=8) alone, fusion + fusion_bwd(B=8) alone, image_bwd(B=8) alone
    b)  image_fwd(B=8) on stream A  ||  fusion + fusion_bwd(B=8) on stream B
    c)  image_bwd(B=8) on stream A  ||  fusion + fusion_bwd(B=8) on stream B
  and the B = 16 phases alone (what the step runs today).

python tools/microbatch_probe.py [--reps 20]   -> one JSON line
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    import bench
    from bridgeqa_amd import fusion_ops
    from bridgeqa_amd.pipeline import PhasedTrainStep
    dev = torch.device("cuda", 0)
    fusion_ops.set_compute_dtype(torch.bfloat16)
    torch.manual_seed(0)

    class A(object):
        points, cin, image, batch = 40000, 132, 512, 16
    model = bench.build_model("c3", A.cin, A.image).to(dev)

    def pipe_for(B, seed):
        batch = bench.make_batch(A, "c3", B, seed, dev)
        p = PhasedTrainStep(model, batch, bench.det_loss, bench.fusion_loss, optimizer=None, use_graphs=True, next_batch=batch)
        p.capture(warmup=2)
        for _ in range(2):
            p.step()
        torch.cuda.synchronize()
        return p

    sa, sb = torch.cuda.Stream(device=dev, priority=-1), torch.cuda.Stream(device=dev, priority=0)

    def timed(plan):
        """plan: {stream: [graphs]} -> per-stream ms (median over reps), both streams released together"""
        res = {k: [] for k in plan}
        for _ in range(a.reps + 3):
            torch.cuda.synchronize()
            evs = {}
            gate = torch.cuda.Event()
            gate.record(torch.cuda.current_stream(dev))
            for s_, graphs in plan.items():
                s_.wait_event(gate)
                with torch.cuda.stream(s_):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(s_)
                    for g in graphs:
                        g.replay()
                    e1.record(s_)
                    evs[s_] = (e0, e1)
            torch.cuda.synchronize()
            for s_, (e0, e1) in evs.items():
                res[s_].append(e0.elapsed_time(e1))
        med = lambda v: sorted(v[3:])[len(v[3:]) // 2]
        return {k: round(med(v), 3) for k, v in res.items()}

    out = {}
    p16 = pipe_for(16, 42)
    g = p16.graphs
    out["B16_alone"] = {"image_fwd": timed({sa: [g["image_fwd"]]})[sa], "fusion": timed({sa: [g["fusion"], g["fusion_bwd"]]})[sa],
                        "image_bwd": timed({sa: [g["image_bwd"]]})[sa]}
    p16.graphs = None
    del p16, g
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    p1, p2 = pipe_for(8, 42), pipe_for(8, 43)
    g1, g2 = p1.graphs, p2.graphs
    out["B8_alone"] = {"image_fwd": timed({sa: [g1["image_fwd"]]})[sa], "fusion": timed({sb: [g2["fusion"], g2["fusion_bwd"]]})[sb],
                       "image_bwd": timed({sa: [g1["image_bwd"]]})[sa]}
    r = timed({sa: [g1["image_fwd"]], sb: [g2["fusion"], g2["fusion_bwd"]]})
    out["B8_image_fwd_beside_fusion"] = {"image_fwd": r[sa], "fusion": r[sb]}
    r = timed({sa: [g1["image_bwd"]], sb: [g2["fusion"], g2["fusion_bwd"]]})
    out["B8_image_bwd_beside_fusion"] = {"image_bwd": r[sa], "fusion": r[sb]}
    # priorities swapped: the fusion chain on the high-priority stream
    r = timed({sb: [g1["image_fwd"]], sa: [g2["fusion"], g2["fusion_bwd"]]})
    out["B8_image_fwd_beside_fusion_fusion_high_prio"] = {"image_fwd": r[sb], "fusion": r[sa]}
    r = timed({sb: [g1["image_bwd"]], sa: [g2["fusion"], g2["fusion_bwd"]]})
    out["B8_image_bwd_beside_fusion_fusion_high_prio"] = {"image_bwd": r[sb], "fusion": r[sa]}
    b16, b8 = out["B16_alone"], out["B8_alone"]
    serial_now = b16["image_fwd"] + b16["fusion"] + b16["image_bwd"]
    f1 = out["B8_image_fwd_beside_fusion"]
    f2 = out["B8_image_bwd_beside_fusion"]
    piped = b8["image_fwd"] + max(f1["image_fwd"], f1["fusion"]) + max(f2["image_bwd"], f2["fusion"]) + b8["image_bwd"]
    out["chain_ms"] = {"today_B16_serial": round(serial_now, 3), "two_microbatches_predicted": round(piped, 3),
                       "predicted_gain": round(serial_now - piped, 3),
                       "note": "image_fwd(mb0) -> [fusion(mb0) || image_fwd(mb1)] -> [fusion(mb1) || image_bwd(mb0)] -> image_bwd(mb1); "
                               "the detector's phases (other stream) and the weight-gradient flushes inside the phases are as captured"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()

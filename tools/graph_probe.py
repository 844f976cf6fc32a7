"""Which part of the c3 step survives HIP-graph capture?  Each candidate runs in its own process."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def child(which):
    import torch
    from bridgeqa_amd import fusion_ops
    fusion_ops.set_compute_dtype(torch.bfloat16)
    import bench
    dev = torch.device("cuda")
    torch.manual_seed(0)
    sys.argv = ["bench.py"]
    args = bench.parse()
    model = bench.build_model("c3", 132, 512).to(dev)
    args.cin = 132
    batch = bench.make_batch(args, "c3", 16, 42, dev)
    blip = model.blip_model
    B = 16
    img_embeds = torch.randn(B, 1025, 768, device=dev)
    obj = torch.randn(B, 256, 256, device=dev)
    om = torch.ones(B, 256, dtype=torch.long, device=dev)
    def f_vit():
        return blip.visual_encoder(batch["images"][:, 0]).float().square().mean()
    def f_text():
        loss, fused, _ = blip(None, batch["question"], batch["answer"], image_embeds=img_embeds,
                              scene_object_embeds=obj, scene_object_mask=om, data_dict={})
        return loss
    def f_det():
        return bench.det_loss(model.detect({k: v for k, v in batch.items() if k not in ("images", "question", "answer")}))
    def f_full():
        return bench.total_loss(model(dict(batch)))
    f = {"vit": f_vit, "text": f_text, "det": f_det, "full": f_full, "fullopt": f_full}[which]
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, fused=True, capturable=True) if which == "fullopt" else None
    params = [p for p in model.parameters()]
    def step():
        for p in params:
            if p.grad is not None: p.grad.zero_()
        l = f(); l.backward()
        if opt is not None: opt.step()
        return l
    import time
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): step()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): step()
    torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 5
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        l = step()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): g.replay()
    torch.cuda.synchronize(); rep = (time.perf_counter() - t0) / 5
    print("%s: capture OK  eager %.1f ms  replay %.1f ms  loss %.4f" % (which, eager * 1e3, rep * 1e3, l.item()), flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        for w in ("det", "vit", "text", "full"):
            r = subprocess.run([sys.executable, __file__, w], capture_output=True, text=True)
            tail = [l for l in (r.stdout + r.stderr).splitlines() if "Warning" not in l and "AccumulateGrad" not in l and "run_backward" not in l and "libdrm" not in l]
            print(w, "rc", r.returncode, "|", " / ".join(tail[-3:]))

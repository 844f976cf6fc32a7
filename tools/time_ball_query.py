"""SA1's ball query alone (16 scenes x 40000 points uniform in an 8 x 8 x 3 m box -> 2048 FPS centres, radius 0.2, 64 samples:
balls hold < 64 points, i.e. every centre scans the whole scene -- also the common case on real surfaces)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bridgeqa_amd import _ext  # noqa: E402

torch.manual_seed(0)
xyz = torch.rand(16, 40000, 3, device="cuda") * torch.tensor([8.0, 8.0, 3.0], device="cuda")
inds = _ext.furthest_point_sampling(xyz, 2048)
new_xyz = torch.gather(xyz, 1, inds.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
for _ in range(2):
    _ext.ball_query(new_xyz, xyz, 0.2, 64)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    idx = _ext.ball_query(new_xyz, xyz, 0.2, 64)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 100
print("ball query SA1 (16 x 40000 -> 2048 x 64): %.1f us  = %.2f TB/s of point coordinates read through L2" % (us, 16 * 2048 * 40000 * 12 / us / 1e6))
# round 6: the one-workgroup-per-scene build of round 5 against the multi-workgroup build (default)
# (replayed from a HIP graph, as inside the step: launched one by one from Python the five launches of mode 1 are host-bound)
for mode in (0, 1):
    _ext.ball_query_grid_build_mode(mode)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(2):
            _ext.ball_query(new_xyz, xyz, 0.2, 64)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        for _ in range(10):
            idx_m = _ext.ball_query(new_xyz, xyz, 0.2, 64)
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print("grid build mode %d (%s): build + query %.1f us under graph replay; identical indices: %s"
          % (mode, "one workgroup per scene" if mode == 0 else "box, cell ids, 16 cell-chunk workgroups per scene", e0.elapsed_time(e1) * 10,
             bool(torch.equal(idx, idx_m))))
    del g
# round 5: the same query with the grid withdrawn (exhaustive scan), and the two launches of the grid path apart
_ext.BALL_QUERY_GRID_MIN_N[0] = 1 << 30
for _ in range(2):
    _ext.ball_query(new_xyz, xyz, 0.2, 64)
torch.cuda.synchronize()
e0.record()
for _ in range(10):
    idx2 = _ext.ball_query(new_xyz, xyz, 0.2, 64)
e1.record()
torch.cuda.synchronize()
print("exhaustive scan: %.1f us; identical indices: %s" % (e0.elapsed_time(e1) * 100, bool(torch.equal(idx, idx2))))

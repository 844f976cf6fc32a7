"""PARITY TESTS PROPER: the HIP operators (through the C ABI) against the CPU oracle on the same
seeded inputs, against the committed golden vectors, and -- at BASELINE.json's full sizes --
against size-independent properties.  Index work is compared bit-exactly; fp32 copies bit-exactly;
atomics-based gradients with a stated tolerance."""
import numpy as np
import pytest
import torch

from conftest import scene
from test_modules_cpu import run_ops_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ext():
    from bridgeqa_amd import _ext
    return _ext


def test_single_hip_runtime_loaded(ext, dev):
    torch.zeros(1, device=dev)
    libs = {l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l}
    assert len(libs) == 1, libs  # libbqhip.so must share torch's HIP runtime (streams are passed across)
    assert any("libbqhip.so" in l for l in open("/proc/self/maps"))


FPS_CASES = [(1, 1, 1), (1, 2, 2), (2, 7, 7), (2, 63, 20), (2, 64, 64), (2, 65, 33), (3, 100, 100), (2, 255, 64),
             (2, 256, 100), (2, 257, 100), (2, 511, 128), (2, 512, 256), (2, 513, 256), (2, 1000, 300),
             (2, 1024, 512), (2, 2048, 1024), (2, 3000, 500), (2, 4096, 1024), (1, 8192, 700), (1, 10000, 512),
             (1, 16384, 300), (1, 20000, 256), (1, 24576, 200), (1, 24577, 200), (1, 30000, 128)]


@pytest.mark.parametrize("B,N,m", FPS_CASES)
def test_fps_index_exact_vs_oracle(ext, oracle, dev, B, N, m):
    xyz = scene(B, N, 0, seed=N * 7 + m)
    got = ext.furthest_point_sampling(xyz.to(dev), m).cpu()
    assert torch.equal(got, oracle.furthest_point_sampling(xyz, m))


def test_fps_bucketed_kernel_edge_cases(ext, oracle, dev):
    """The pruned kernel (N > 4096) on inputs built to break a pruning or tie rule."""
    g = torch.Generator().manual_seed(5)
    cases = {}
    # exact ties everywhere: lattice of 32x16x10 = 5120 points
    lat = np.stack(np.meshgrid(np.arange(32) * 0.25, np.arange(16) * 0.25, np.arange(10) * 0.25, indexing="ij"), -1)
    cases["lattice"] = (torch.tensor(lat.reshape(1, -1, 3) + 0.5, dtype=torch.float32), 600)
    # every point identical / two clusters of duplicates
    same = torch.full((1, 5000, 3), 1.25)
    cases["identical"] = (same, 40)
    dup = scene(1, 6000, 0, 1)
    dup[0, 1000:3000] = dup[0, 7]
    dup[0, 3000:5000] = dup[0, 11]
    cases["duplicates"] = (dup, 4200)  # m beyond the number of distinct points
    # dense blob + far outliers (degenerate grid), and a scene far from the origin
    blob = torch.randn(1, 9000, 3, generator=g) * 0.01 + 3.0
    blob[0, ::1000] += 100.0
    cases["blob+outliers"] = (blob, 300)
    cases["translated"] = (scene(1, 7000, 0, 2) + 1000.0, 300)
    # many points inside the origin skip ball, first point among them
    ob = scene(2, 5000, 0, 3)
    ob[:, :2500] *= 0.002
    cases["origin ball"] = (ob, 2600)
    cases["all skipped"] = (torch.zeros(1, 4500, 3), 10)
    # flat (z constant) and line-like scenes
    flat = scene(1, 8000, 0, 4); flat[..., 2] = 1.0
    cases["flat"] = (flat, 400)
    line = scene(1, 8000, 0, 5); line[..., 1:] = 2.0
    cases["line"] = (line, 400)
    # minima kept outside LDS (N > 40448) and two rows per lane (N > 65536)
    cases["N=41000"] = (scene(1, 41000, 0, 6), 300)
    cases["N=80000"] = (scene(2, 80000, 0, 7), 256)
    for name, (xyz, m) in cases.items():
        xyz = xyz.contiguous()
        got = ext.furthest_point_sampling(xyz.to(dev), m).cpu()
        assert torch.equal(got, oracle.furthest_point_sampling(xyz, m)), name
        assert torch.equal(got, ext.furthest_point_sampling_bruteforce(xyz.to(dev), m).cpu()), name


def test_fps_edge_cases_vs_oracle(ext, oracle, dev):
    # exact ties on a lattice (tie order = bit-reversed reference-thread id, then k), several block sizes
    for nx, ny, nz in ((4, 4, 2), (8, 8, 4), (16, 8, 5), (16, 16, 8)):
        g = np.stack(np.meshgrid(np.arange(nx) * 0.5, np.arange(ny) * 0.5, np.arange(nz) * 0.5, indexing="ij"), -1)
        xyz = torch.tensor(g.reshape(1, -1, 3) + 1.0, dtype=torch.float32).contiguous()
        m = xyz.shape[1]
        assert torch.equal(ext.furthest_point_sampling(xyz.to(dev), m).cpu(), oracle.furthest_point_sampling(xyz, m))
    # duplicates + origin-ball points + m > number of distinct points
    xyz = scene(2, 700, 0, 3)
    xyz[0, 100:140] = xyz[0, 5]
    xyz[0, 0] = 0.0
    xyz[0, 9] = torch.tensor([0.03, 0.0, 0.0])
    xyz[0, 10] = torch.tensor([0.0316, 0.0, 0.0])
    xyz[0, 11] = torch.tensor([0.03163, 0.0, 0.0])
    xyz[1, :] = xyz[1, :1]  # every point identical
    assert torch.equal(ext.furthest_point_sampling(xyz.to(dev), 700).cpu(), oracle.furthest_point_sampling(xyz, 700))
    z = torch.zeros(1, 50, 3)  # everything inside the skip ball
    assert ext.furthest_point_sampling(z.to(dev), 8).cpu().tolist() == [[0] * 8]
    tiny = scene(1, 5, 0, 12)  # m > N
    assert torch.equal(ext.furthest_point_sampling(tiny.to(dev), 9).cpu(), oracle.furthest_point_sampling(tiny, 9))
    assert ext.furthest_point_sampling(torch.zeros(0, 4, 3, device=dev), 2).shape == (0, 2)


@pytest.mark.parametrize("B,N,M,r,S", [(2, 50, 10, 0.5, 4), (2, 1000, 64, 0.4, 16), (2, 1000, 257, 3.0, 32),
                                       (1, 4096, 512, 0.2, 64), (2, 300, 300, 0.05, 8), (1, 130, 3, 100.0, 64),
                                       (1, 64, 64, 1.0, 5), (1, 5000, 100, 1.2, 16)])
def test_ball_query_index_exact_vs_oracle(ext, oracle, dev, B, N, M, r, S):
    xyz = scene(B, N, 0, seed=N + M)
    new_xyz = xyz[:, torch.randperm(N, generator=torch.Generator().manual_seed(1))[:M]].contiguous()
    if r < 0.1:
        new_xyz[:, M // 2:] += 50.0  # empty balls -> all-zero rows
    got = ext.ball_query(new_xyz.to(dev), xyz.to(dev), r, S).cpu()
    assert torch.equal(got, oracle.ball_query(new_xyz, xyz, r, S))


def test_gather_group_interp_vs_oracle(ext, oracle, dev):
    g = torch.Generator().manual_seed(0)
    for (B, C, N, M, S) in ((2, 5, 300, 40, 8), (1, 131, 2048, 1024, 32), (2, 3, 777, 33, 5), (1, 1, 64, 64, 64)):
        pts = torch.randn(B, C, N, generator=g)
        idx = torch.randint(0, N, (B, M, S), generator=g, dtype=torch.int32)
        idx[:, :, S // 2:] = idx[:, :, :1]  # padded neighbourhoods: repeated ids
        assert torch.equal(ext.group_points(pts.to(dev), idx.to(dev)).cpu(), oracle.group_points(pts, idx))
        go = torch.randn(B, C, M, S, generator=g)
        torch.testing.assert_close(ext.group_points_grad(go.to(dev), idx.to(dev), N).cpu(),
                                   oracle.group_points_grad(go, idx, N), rtol=1e-4, atol=1e-4)  # fp32 atomics: order
        gi = torch.randint(0, N, (B, M), generator=g, dtype=torch.int32)
        assert torch.equal(ext.gather_points(pts.to(dev), gi.to(dev)).cpu(), oracle.gather_points(pts, gi))
        gg = torch.randn(B, C, M, generator=g)
        torch.testing.assert_close(ext.gather_points_grad(gg.to(dev), gi.to(dev), N).cpu(),
                                   oracle.gather_points_grad(gg, gi, N), rtol=1e-5, atol=1e-5)
    for (B, n, m, C) in ((2, 120, 40, 6), (1, 1024, 512, 256), (2, 300, 2, 4), (1, 2500, 1500, 3)):
        unknown, known = scene(B, n, 0, 1), scene(B, m, 0, 2)
        k = min(5, m)
        known[:, :k] = unknown[:, :k]
        d2, idx = ext.three_nn(unknown.to(dev), known.to(dev))
        od2, oidx = oracle.three_nn(unknown, known)
        assert torch.equal(idx.cpu(), oidx) and torch.equal(d2.cpu(), od2)
        dist, idx_b = ext.three_nn_dist(unknown.to(dev), known.to(dev))  # fused correctly-rounded sqrt
        assert torch.equal(idx_b.cpu(), oidx)
        np.testing.assert_array_equal(dist.cpu().numpy(), np.sqrt(od2.numpy()))
        w = torch.rand(B, n, 3, generator=g)
        feats = torch.randn(B, C, m, generator=g)
        assert torch.equal(ext.three_interpolate(feats.to(dev), idx, w.to(dev)).cpu(),
                           oracle.three_interpolate(feats, oidx, w))  # same op order, no contraction => exact
        go = torch.randn(B, C, n, generator=g)
        torch.testing.assert_close(ext.three_interpolate_grad(go.to(dev), idx, w.to(dev), m).cpu(),
                                   oracle.three_interpolate_grad(go, oidx, w, m), rtol=1e-4, atol=1e-4)


def test_group_concat_equals_reference_composition(ext, oracle, dev):
    g = torch.Generator().manual_seed(3)
    for (B, C, N, M, S, r, norm) in ((2, 5, 512, 64, 16, 0.9, True), (1, 0, 300, 20, 8, 0.3, True),
                                     (1, 128, 2048, 1024, 32, 0.4, True), (2, 4, 100, 10, 4, 0.2, False)):
        xyz = scene(B, N, 0, 5)
        new_xyz = xyz[:, :M].contiguous()
        idx = oracle.ball_query(new_xyz, xyz, r, S)
        feats = torch.randn(B, C, N, generator=g) if C else None
        # reference composition (pointnet2_utils.py:348-359) on CPU
        gx = oracle.group_points(xyz.transpose(1, 2).contiguous(), idx) - new_xyz.transpose(1, 2).unsqueeze(-1)
        if norm:
            gx = gx / r
        want = torch.cat([gx, oracle.group_points(feats, idx)], 1) if C else gx
        got = ext.group_concat(xyz.to(dev), new_xyz.to(dev), feats.to(dev) if C else None, idx.to(dev), r, norm)
        assert torch.equal(got.cpu(), want)
        go = torch.randn(B, C + 3, M, S, generator=g)
        gf, gxyz, gnew = ext.group_concat_grad(go.to(dev), idx.to(dev), N, r, norm, True, True, True)
        s = go[:, :3] / r if norm else go[:, :3]
        want_xyz = oracle.group_points_grad(s.contiguous(), idx, N).transpose(1, 2)
        torch.testing.assert_close(gxyz.cpu(), want_xyz, rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(gnew.cpu(), -s.sum(-1).transpose(1, 2), rtol=1e-4, atol=1e-4)
        if C:
            torch.testing.assert_close(gf.cpu(), oracle.group_points_grad(go[:, 3:].contiguous(), idx, N),
                                       rtol=1e-4, atol=1e-4)


def test_autograd_ops_vs_reference_golden_on_gpu(ext, golden, dev):
    from bridgeqa_amd import pointnet2_utils as pu
    assert pu.backend() is ext  # the HIP backend, not a substitute
    run_ops_golden(golden("pn2_ops.npz"), pu, lambda a: torch.from_numpy(a).to(dev))


def test_full_size_c2_properties_and_oracle_spot_check(ext, oracle, dev):
    """BASELINE config 2 sizes: B=16, N=40000 -> 2048 centres, r=0.2, S=64."""
    B, N, M = 16, 40000, 2048
    xyz = scene(B, N, 0, seed=42)
    dxyz = xyz.to(dev)
    inds = ext.furthest_point_sampling(dxyz, M)
    ci = inds.cpu()
    assert (ci[:, 0] == 0).all() and ((ci >= 0) & (ci < N)).all()
    assert all(len(set(r.tolist())) == M for r in ci)  # distinct points -> unique ids
    assert torch.equal(ext.furthest_point_sampling(dxyz, 256).cpu(), ci[:, :256])  # prefix property
    # permuting scenes permutes results (no cross-scene state)
    perm = torch.tensor([3, 0, 15, 7])
    assert torch.equal(ext.furthest_point_sampling(dxyz[perm.to(dev)].contiguous(), 512).cpu(), ci[perm, :512])
    assert torch.equal(ci[:2], oracle.furthest_point_sampling(xyz[:2].contiguous(), M))  # full size, 2 scenes
    new_xyz = ext.gather_points(dxyz.transpose(1, 2).contiguous(), inds).transpose(1, 2).contiguous()
    assert torch.equal(new_xyz.cpu(), torch.gather(xyz, 1, ci.long()[..., None].expand(-1, -1, 3)))
    idx = ext.ball_query(new_xyz, dxyz, 0.2, 64).cpu()
    assert (idx == ci[..., None]).any(-1).all()  # a centre is inside its own ball (d=0) and balls hold < 64 points here
    # ids ascending up to the pad, pads repeat the first hit
    d = idx[:, :, 1:] - idx[:, :, :-1]
    assert ((d > 0) | (idx[:, :, 1:] == idx[:, :, :1])).all()
    assert torch.equal(idx[:2], oracle.ball_query(new_xyz[:2].cpu().contiguous(), xyz[:2].contiguous(), 0.2, 64))
    # SA2-level FPS on the FPS-ordered set is the identity prefix (backbone_module.py:111) EXCEPT where two
    # candidates tie exactly in fp32 (then the index-dependent tie rule differs between the two orderings; the
    # oracle shows the same 2 swapped ids in scene 1 of this seed) => exact vs oracle, near-identity as a property
    inds2 = ext.furthest_point_sampling(new_xyz, 1024).cpu()
    assert torch.equal(inds2, oracle.furthest_point_sampling(new_xyz.cpu().contiguous(), 1024))
    assert (inds2 == torch.arange(1024, dtype=torch.int32).expand(B, -1)).float().mean() > 0.999


def _bf16_ulp_diff(a, b):
    """distance in bf16 units-in-the-last-place between two bf16 tensors (sign-magnitude -> ordered integers)"""
    ia = a.contiguous().view(torch.int16).to(torch.int32)
    ib = b.contiguous().view(torch.int16).to(torch.int32)
    oa = torch.where(ia < 0, -(ia & 0x7FFF), ia)
    ob = torch.where(ib < 0, -(ib & 0x7FFF), ib)
    return (oa - ob).abs().max().item()


def test_point_major_and_bf16_grouping_vs_oracle_composition(ext, oracle, dev):
    """bq_group_concat_pm / bq_group_concat_pm_grad / bq_group_concat_bf16 (the forms the c3 bench runs) against the
    reference composition (lib/pointnet2/pointnet2_utils.py:348-359) over the CPU oracle: fp32 output bit-exact (up
    to the layout permutation), bf16 output <= 1 bf16 ulp from the rounded fp32 composition, gradients 1e-4 (atomics)."""
    g = torch.Generator().manual_seed(11)
    for (B, C, N, M, S, r, norm) in ((2, 5, 512, 64, 16, 0.9, True), (1, 0, 300, 20, 8, 0.3, True),
                                     (2, 132, 4096, 256, 64, 0.4, True), (1, 128, 2048, 1024, 32, 0.4, True),
                                     (2, 4, 100, 10, 4, 0.2, False)):
        xyz = scene(B, N, 0, 5)
        new_xyz = xyz[:, :M].contiguous()
        idx = oracle.ball_query(new_xyz, xyz, r, S)
        feats = torch.randn(B, C, N, generator=g) if C else None
        gx = oracle.group_points(xyz.transpose(1, 2).contiguous(), idx) - new_xyz.transpose(1, 2).unsqueeze(-1)
        if norm:
            gx = gx / r
        want = torch.cat([gx, oracle.group_points(feats, idx)], 1) if C else gx          # (B, 3+C, M, S)
        want_pm = want.permute(0, 2, 3, 1).contiguous()                                     # (B, M, S, 3+C)
        # point-major source rows: the interleaved (B, N, 3+C) cloud read in place (a strided view, as the bench does)
        cloud = torch.cat([xyz, feats.transpose(1, 2)], -1).contiguous().to(dev) if C else None
        feats_pm = cloud[:, :, 3:] if C else None
        dxyz, dnew, didx = xyz.to(dev), new_xyz.to(dev), idx.to(dev)
        got32 = ext.group_concat_pm(dxyz, dnew, feats_pm, didx, r, norm, torch.float32)
        assert torch.equal(got32.cpu(), want_pm)
        got16 = ext.group_concat_pm(dxyz, dnew, feats_pm, didx, r, norm, torch.bfloat16)
        assert _bf16_ulp_diff(got16.cpu(), want_pm.to(torch.bfloat16)) <= 1
        got16c = ext.group_concat(dxyz, dnew, feats.to(dev) if C else None, didx, r, norm, torch.bfloat16)
        assert _bf16_ulp_diff(got16c.cpu(), want.to(torch.bfloat16)) <= 1
        # gradients: point-major grad_out (B, M, S, 3+C), fp32 and bf16
        go = torch.randn(B, M, S, C + 3, generator=g)
        for dt, tol in ((torch.float32, 1e-4), (torch.bfloat16, 1e-4)):
            go_d = go.to(dt)
            go_ref = go_d.float().permute(0, 3, 1, 2).contiguous()                          # (B, 3+C, M, S)
            gf, gxyz, gnew = ext.group_concat_pm_grad(go_d.to(dev), didx, N, r, norm, True, True, True)
            s = go_ref[:, :3] / r if norm else go_ref[:, :3]
            want_xyz = oracle.group_points_grad(s.contiguous(), idx, N).transpose(1, 2)
            torch.testing.assert_close(gxyz.cpu(), want_xyz, rtol=tol, atol=tol)
            torch.testing.assert_close(gnew.cpu(), -s.sum(-1).transpose(1, 2), rtol=tol, atol=tol)
            if C:
                want_f = oracle.group_points_grad(go_ref[:, 3:].contiguous(), idx, N).transpose(1, 2)  # (B, N, C)
                torch.testing.assert_close(gf.cpu(), want_f, rtol=tol, atol=tol)


@pytest.mark.gpu
def test_ball_query_background_grid_equals_the_full_grid(ext, dev):
    """bq_ball_query_background (about one workgroup per CU, grid-stride over the centre pairs): the same indices as
    bq_ball_query at SA1's size, with an odd centre count and batch sizes on both sides of the CU count"""
    torch.manual_seed(3)
    for B, N, M, r, S in ((16, 40000, 2048, 0.2, 64), (3, 5000, 777, 0.4, 32), (300, 200, 9, 0.5, 16)):
        xyz = torch.rand(B, N, 3, device=dev) * 4.0
        new_xyz = xyz[:, torch.randperm(N, device=dev)[:M]].contiguous()
        assert torch.equal(ext.ball_query(new_xyz, xyz, r, S, background=True), ext.ball_query(new_xyz, xyz, r, S))


def _grid_vs_scan(ext, new_xyz, xyz, r, S):
    """the same call through the grid (csrc/ball_query_grid.hip) and through the exhaustive scan"""
    prev = ext.BALL_QUERY_GRID_MIN_N[0]
    try:
        ext.BALL_QUERY_GRID_MIN_N[0] = 1
        a = ext.ball_query(new_xyz, xyz, r, S)
        # (round 6) both ways of binning the scene -- box / cell ids / 16 cell-chunk workgroups per scene in three launches
        # (default from N = 4096) and one workgroup per scene -- must give the same indices: the order of the records inside
        # a cell never reaches the output
        mode = ext.ball_query_grid_build_mode(0)
        try:
            a1 = ext.ball_query(new_xyz, xyz, r, S)
        finally:
            ext.ball_query_grid_build_mode(mode)
        assert torch.equal(a, a1)
        ext.BALL_QUERY_GRID_MIN_N[0] = 1 << 30
        b = ext.ball_query(new_xyz, xyz, r, S)
    finally:
        ext.BALL_QUERY_GRID_MIN_N[0] = prev
    return a, b


@pytest.mark.parametrize("B,N,M,r,S", [(2, 8192, 512, 0.2, 64), (3, 40000, 2048, 0.2, 64), (1, 8193, 100, 0.05, 16),
                                       (2, 20000, 777, 0.4, 32), (1, 40000, 256, 3.0, 64), (2, 9000, 300, 1.2, 16),
                                       (1, 100, 7, 0.5, 8), (1, 80000, 2048, 0.2, 64)])
def test_grid_ball_query_index_exact(ext, oracle, dev, B, N, M, r, S):
    """bq_ball_query_grid against the oracle (ball_query_gpu.cu:9-44 restated) and against the exhaustive kernel: identical
    indices -- few-hit balls (ranked in LDS), empty balls, centres far outside the scene's bounding box, radii from a
    fraction of a cell to the whole scene (the grid collapses to a few cells and every ball overflows into the fallback scan)"""
    xyz = scene(B, N, 0, seed=N + M)
    new_xyz = xyz[:, torch.randperm(N, generator=torch.Generator().manual_seed(1))[:M]].contiguous()
    new_xyz[:, : M // 8] += 0.37           # centres that are not points
    new_xyz[:, M // 8: M // 4] += 50.0     # far outside: empty balls, all-zero rows
    new_xyz[:, M // 4: M // 4 + 3] -= 7.0
    got, scan = _grid_vs_scan(ext, new_xyz.to(dev), xyz.to(dev), r, S)
    assert torch.equal(got, scan)
    if N * M <= 2e8:
        assert torch.equal(got.cpu(), oracle.ball_query(new_xyz, xyz, r, S))


def test_grid_ball_query_dense_clusters_ties_and_degenerate_scenes(ext, oracle, dev):
    """what a uniform random scene does not exercise: thousands of points inside one ball (more hits than the LDS list
    holds: the per-centre fallback), hit counts on both sides of nsample and of the list capacity, points exactly ON the
    sphere (d2 == r^2 is outside: strict '<'), duplicated points, a planar scene (zero extent along z), all points equal"""
    g = torch.Generator().manual_seed(3)
    N = 12000
    base = torch.rand(1, N, 3, generator=g) * torch.tensor([6.0, 6.0, 2.5])
    xyz = base.clone()
    xyz[0, :3000] = torch.tensor([1.0, 1.0, 1.0]) + 0.05 * torch.randn(3000, 3, generator=g)      # 3000 points in one ball
    xyz[0, 3000:3300] = torch.tensor([4.0, 4.0, 1.0]) + 0.08 * torch.randn(300, 3, generator=g)   # ~256 hits: the list's edge
    xyz[0, 3300:3360] = torch.tensor([2.0, 5.0, 1.0]) + 0.05 * torch.randn(60, 3, generator=g)    # ~60 hits: nsample's edge
    xyz[0, 5000:5100] = xyz[0, 4000:4100]                                                           # duplicates
    ctr = torch.cat([xyz[:, [10, 3010, 3310, 4000, 5000, 7777]], torch.tensor([[[1.0, 1.0, 1.0], [4.0, 4.0, 1.0], [2.0, 5.0, 1.0]]])], 1)
    # points exactly on the sphere of the last centre: offsets whose squared length is exactly r^2 = 0.25 in fp32
    c = torch.tensor([3.0, 2.0, 0.5])
    xyz[0, 6000] = c + torch.tensor([0.5, 0.0, 0.0]); xyz[0, 6001] = c + torch.tensor([0.3, 0.4, 0.0]); xyz[0, 6002] = c - torch.tensor([0.0, 0.3, 0.4])
    ctr = torch.cat([ctr, c.view(1, 1, 3)], 1).contiguous()
    for r, S in ((0.2, 64), (0.5, 64), (0.2, 8), (0.5, 300)):
        got, scan = _grid_vs_scan(ext, ctr.to(dev), xyz.to(dev), r, S)
        assert torch.equal(got, scan), (r, S)
        assert torch.equal(got.cpu(), oracle.ball_query(ctr, xyz.contiguous(), r, S)), (r, S)
    flat = base.clone(); flat[..., 2] = 0.7
    same = torch.full((1, N, 3), 1.5)
    for pts in (flat, same):
        q = pts[:, :50].contiguous() + 0.01
        got, scan = _grid_vs_scan(ext, q.to(dev), pts.to(dev), 0.2, 16)
        assert torch.equal(got, scan) and torch.equal(got.cpu(), oracle.ball_query(q, pts.contiguous(), 0.2, 16))


def test_inverted_index_and_deterministic_scatter_gradients(ext, oracle, dev):
    """csrc/invert.hip: bq_invert_index is a CSR of the positions naming every point, ascending inside a
    point; the feature gradient of the point-major grouping and the gradient of three_interpolate as GATHERS over it equal
    the oracle's (group_points_gpu.cu:43-75, interpolate_gpu.cu:116-154 restated) and are bitwise reproducible -- the atomic
    scatters are not"""
    g = torch.Generator().manual_seed(5)
    B, N, M, S, C = 3, 5000, 256, 32, 61
    idx = torch.randint(0, N, (B, M, S), generator=g, dtype=torch.int32)
    idx[:, :, S // 2:] = idx[:, :, :1]                      # the padding pattern of a sparse ball: one index repeated
    idx[0, :40] = 7                                         # one point named by 1280 slots
    d_idx = idx.to(dev)
    start, slots = ext.invert_index(d_idx, N)
    start, slots = start.cpu().long(), slots.cpu().long()
    flat = idx.view(B, -1).long()
    assert start[0] == 0 and start[-1] == B * M * S and (start[1:] >= start[:-1]).all()
    for b, v in ((0, 7), (1, int(flat[1, 5])), (2, int(flat[2, -1])), (2, N - 1)):
        got = slots[start[b * N + v]:start[b * N + v + 1]]
        want = (flat[b] == v).nonzero().flatten() + b * M * S
        assert torch.equal(got, want), (b, v)
    # the whole table against a stable argsort
    order = np.argsort((flat + torch.arange(B)[:, None] * N).flatten().numpy(), kind="stable")
    assert np.array_equal(slots.numpy(), order)
    # feature gradient of group_concat_pm: padded bf16 rows (ld = 64) and contiguous fp32 rows
    for dt, ld in ((torch.bfloat16, 64), (torch.float32, 64)):
        go = torch.zeros(B, M, S, ld, dtype=dt, device=dev)
        go[..., :3 + C] = torch.randn(B, M, S, 3 + C, generator=g).to(dev).to(dt)
        rows = go[..., :3 + C]
        inv = ext.invert_index(d_idx, N)
        outs = [ext.group_concat_pm_grad_gather(rows, inv, N, 0.2, True, True, True, True) for _ in range(3)]
        for k in range(3):   # features, point coordinates, centres
            assert outs[0][k] is not None and torch.equal(outs[0][k], outs[1][k]) and torch.equal(outs[0][k], outs[2][k])
        want = oracle.group_points_grad(rows.float().cpu().permute(0, 3, 1, 2)[:, 3:].contiguous(), idx, N).transpose(1, 2)
        torch.testing.assert_close(outs[0][0].cpu(), want, rtol=2e-5, atol=2e-4)
        atomic = ext.group_concat_pm_grad(rows, d_idx, N, 0.2, True, True, True, True)
        for k in range(3):
            torch.testing.assert_close(outs[0][k], atomic[k], rtol=2e-5, atol=2e-3)
    # three_interpolate: n unknown points over m known ones
    n, m, Cf = 700, 90, 33
    idx3 = torch.randint(0, m, (B, n, 3), generator=g, dtype=torch.int32)
    w3 = torch.rand(B, n, 3, generator=g)
    go3 = torch.randn(B, Cf, n, generator=g)
    inv3 = ext.invert_index(idx3.to(dev), m)
    got = [ext.three_interpolate_grad_gather(go3.to(dev), inv3, w3.to(dev), m) for _ in range(2)]
    assert torch.equal(got[0], got[1])
    torch.testing.assert_close(got[0].cpu(), oracle.three_interpolate_grad(go3, idx3, w3, m), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("B,L,N", [(1, 1, 1), (2, 2049, 3), (5, 4096, 70000), (32, 2048 * 64, 80000)])
def test_inverted_index_is_a_stable_sort_eager_and_replayed(ext, dev, B, L, N):
    """bq_invert_index at sizes from one pair to configs[5]'s SA1 level (4.2 M pairs over 2.56 M points: three 8-bit passes),
    eagerly and REPLAYED from a HIP graph on fresh indices (the round-5 fault: the first version's library sort died exactly
    there): slots is the stable argsort of scene * N + index, start its CSR"""
    g = torch.Generator().manual_seed(B * 7 + N)
    def draw():
        idx = torch.randint(0, N, (B, L), generator=g, dtype=torch.int32)
        idx[:, L // 2:] = idx[:, :1]
        return idx
    def check(idx, start, slots):
        keys = (idx.long() + torch.arange(B)[:, None] * N).flatten().numpy()
        assert np.array_equal(slots.cpu().numpy(), np.argsort(keys, kind="stable").astype(np.uint32))
        want = np.zeros(B * N + 1, np.int64)
        np.cumsum(np.bincount(keys, minlength=B * N), out=want[1:])
        assert np.array_equal(start.cpu().numpy(), want)
    idx = draw()
    d_idx = idx.to(dev)
    check(idx, *ext.invert_index(d_idx, N))
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        ext.invert_index(d_idx, N)
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        start, slots = ext.invert_index(d_idx, N)
    for _ in range(2):
        idx = draw()
        d_idx.copy_(idx)
        graph.replay()
        torch.cuda.synchronize()
        check(idx, start, slots)

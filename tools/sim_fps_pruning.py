"""Offline model of the bucketed FPS kernel's pruning (numpy): how many 64-point rows does a round touch?
Used to size rows / grid; not part of the product or the tests."""
import sys
import numpy as np

def simulate(N=40000, m=2048, seed=42, ROW=64, NW=16, ppc=32.0, order="snake", skew=False):
    rng = np.random.RandomState(seed)
    xyz = (rng.rand(N, 3) * np.array([8.0, 8.0, 3.0])).astype(np.float32)
    lo, hi = xyz.min(0), xyz.max(0)
    ext = np.maximum(hi - lo, 1e-6)
    target = min(max(N / ppc, 1), 2048)
    e = np.cbrt(ext.prod() / target)
    G = np.clip((ext / e).astype(int), 1, 64)
    while G.prod() > 2048:
        G[np.argmax(G)] -= 1
    c = np.clip(((xyz - lo) * (G / ext)).astype(int), 0, G - 1)
    cx, cy, cz = c[:, 0], c[:, 1], c[:, 2]
    if order == "snake":
        cy = np.where(cz & 1, G[1] - 1 - cy, cy)
        cx = np.where((cz * G[1] + cy) & 1, G[0] - 1 - cx, cx)
    cell = (cz * G[1] + cy) * G[0] + cx
    perm = np.argsort(cell, kind="stable")
    s = xyz[perm]
    nrows = (N + ROW - 1) // ROW
    pad = nrows * ROW - N
    sp = np.concatenate([s, np.full((pad, 3), np.nan, np.float32)]).reshape(nrows, ROW, 3)
    blo, bhi = np.nanmin(sp, 1), np.nanmax(sp, 1)
    md = np.full((nrows, ROW), 1e10, np.float32)
    md[np.isnan(sp[..., 0])] = -1
    rmax = np.full(nrows, 3e38, np.float32)
    p = xyz[0]
    act_hist, wave_hist = [], []
    for j in range(1, m):
        g = np.maximum(np.maximum(blo - p, p - bhi), 0)
        lb = (g * g).sum(1)
        act = np.nonzero(lb < rmax)[0]
        d = ((sp[act] - p) ** 2).sum(-1)
        md[act] = np.fmin(md[act], d)
        rmax[act] = md[act].max(1)
        r = np.argmax(rmax)
        p = sp[r, np.argmax(md[r])]
        act_hist.append(len(act))
        wv = (act + act // NW) % NW if skew else act % NW
        wave_hist.append(np.bincount(wv, minlength=NW).max() if len(act) else 0)
    a, w = np.array(act_hist), np.array(wave_hist)
    print("ROW=%d ppc=%g grid=%s rows=%d | active rows/round: mean %.1f (first 64: %.0f, last 1024: %.1f) | "
          "max rows on one wave: mean %.2f  | total point-updates %.2e vs brute %.2e (%.1fx less)"
          % (ROW, ppc, G.tolist(), nrows, a.mean(), a[:64].mean(), a[-1024:].mean(), w.mean(),
             a.sum() * ROW, N * (m - 1), N * (m - 1) / (a.sum() * ROW)))

if __name__ == "__main__":
    for ROW, ppc, skew in ((64, 32, False), (64, 32, True), (128, 64, False), (128, 64, True)):
        print("skew", skew, end=" ")
        simulate(ROW=ROW, ppc=ppc, skew=skew)

"""Synthetic tie-heavy image content for the pixels -> masks -> bytes checks (round-2 verdict, item 3).

TEST INFRASTRUCTURE ONLY (tests/, bench.py's mask-flip report).  Real images are 8-bit, smooth, flat or blocky: many patches
carry nearly or exactly the same multiset of gray values, the router's thresholds are k-th smallest entropies
(RouterTriple.py:21-34, strict '<'), so the last bits of the entropy maps decide which patches fall under a threshold.
Every family is exactly representable: float32(uint8) / 255.0.
"""
import numpy as np
def families(n=64, H=256, W=256, seed=7):
    """tie-heavy synthetic content: {name: float32 [n,3,H,W] in [0,1]}"""
    rng = np.random.default_rng(seed)
    out = {}
    out["noise8"] = (rng.integers(0, 256, (n, 3, H, W)).astype(np.float32) / 255.0)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    g = np.empty((n, 3, H, W), np.float32)
    for i in range(n):
        a = rng.uniform(0, 2 * np.pi); f = rng.uniform(0.2, 1.0)
        base = 0.5 + 0.45 * np.sin((np.cos(a) * xx + np.sin(a) * yy) * f * 2 * np.pi / max(H, W))
        for c in range(3):
            g[i, c] = base * rng.uniform(0.6, 1.0) + rng.integers(-1, 2, (H, W)) / 255.0
    out["smooth8"] = np.round(np.clip(g, 0, 1) * 255.0).astype(np.float32) / 255.0
    f = np.empty((n, 3, H, W), np.float32)
    for i in range(n):
        img = np.full((3, H, W), rng.integers(0, 256, (3, 1, 1)), np.float32)
        for _ in range(rng.integers(3, 12)):
            y0, x0 = rng.integers(0, H - 8), rng.integers(0, W - 8)
            y1, x1 = rng.integers(y0 + 4, H + 1), rng.integers(x0 + 4, W + 1)
            img[:, y0:y1, x0:x1] = rng.integers(0, 256, (3, 1, 1))
        f[i] = img / 255.0
    out["flat_edges"] = f
    b = rng.integers(0, 256, (n, 3, H // 8, W // 8)).astype(np.float32)
    b = np.repeat(np.repeat(b, 8, axis=2), 8, axis=3)
    ripple = ((xx.astype(np.int64) + yy.astype(np.int64)) % 2).astype(np.float32)[None, None] * rng.integers(0, 3, (n, 3, 1, 1))
    out["blocky8"] = np.clip(b + ripple, 0, 255) / 255.0
    return {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in out.items()}

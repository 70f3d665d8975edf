"""GPU debug: tools/stress_codec.py with the case printed BEFORE it runs (a hang or a mismatch names its case); large grids only"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import control_gic_amd as cg
from oracle import cgic_oracle as orc
g = np.load(os.path.join(ROOT, "tests", "golden", "coders.npz"))
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 30


class _Item:
    def __init__(self, v): self.v = v
    def item(self): return self.v


tables = {}
for name in ("zipf", "big"):
    tables[name] = ({str(int(k)): _Item(float(g[name + "_freq"][int(k)])) for k in g[name + "_order"]}, orc.HuffmanTable(g[name + "_freq"]))
cbk = torch.from_numpy(rng.standard_normal((1024, 4)).astype(np.float32)).cuda()
codecs = {n: cg.GrainCodec(tables[n][0], cbk) for n in tables}
RATIOS = [(0.1, 0.8), (0.1, 0.4), (0.7, 0.3), (0.3, 0.7), (0.0, 0.4), (0.4, 0.0), (1.0, 0.0), (0.0, 1.0), (0.0, 0.0), (0.5, 0.5), (0.33, 0.33)]
t0 = time.time(); n = 0; bad = 0
while time.time() - t0 < budget:
    B = int(rng.integers(1, 6)); h = 4 * int(rng.integers(20, 60)); w = 4 * int(rng.integers(20, 60))
    name = str(rng.choice(list(tables)))
    c, m = RATIOS[int(rng.integers(0, len(RATIOS)))]
    desc = dict(B=B, h=h, w=w, table=name, ratio=(c, m))
    print("case", n, desc, flush=True)
    e16 = torch.from_numpy((rng.random((B, h // 4, w // 4)) * 2.6).astype(np.float32)).cuda()
    e8 = torch.from_numpy((rng.random((B, h // 2, w // 2)) * 2.6).astype(np.float32)).cuda()
    ind = rng.integers(0, 1024, (B, h, w))
    mask, _, _, mode = cg.TripleGrainFixedEntropyRouter(c, m, per_image=True)(e16, e8)
    comp = codecs[name].compress(torch.from_numpy(ind).cuda(), mask, mode)
    torch.cuda.synchronize(); print("  compressed", flush=True)
    host = comp.to_host()
    mks = [t.cpu().numpy() for t in mask]
    for b in range(B):
        ref = orc.compress_image(ind[b], mks[0][b, 0], mks[1][b, 0], mks[2][b, 0], mode, tables[name][1])
        if host[b] != ref:
            bad += 1
            for k, (x, y) in enumerate(zip(host[b], ref)):
                if x != y:
                    x, y = x or b"", y or b""
                    first = next((i for i in range(min(len(x), len(y))) if x[i] != y[i]), min(len(x), len(y)))
                    print(f"  MISMATCH image {b} stream {k} mode {mode}: len {len(x)} vs {len(y)}, first differing byte {first}", flush=True)
    dind, dmask, zq, status = codecs[name].decompress(comp)
    torch.cuda.synchronize(); print("  decompressed", flush=True)
    exp = np.where(mks[2][:, 0] == 1, ind, 0)
    exp = exp + np.repeat(np.repeat(np.where(mks[1][:, 0] == 1, ind[:, ::2, ::2], 0), 2, 1), 2, 2)
    exp = exp + np.repeat(np.repeat(np.where(mks[0][:, 0] == 1, ind[:, ::4, ::4], 0), 4, 1), 4, 2)
    if not (int(status.abs().max()) == 0 and np.array_equal(dind.cpu().numpy(), exp)):
        bad += 1; print("  DECODE MISMATCH status", status.cpu().tolist(), flush=True)
    n += 1
    if bad >= 3: break
print(f"{n} cases, {bad} bad")

"""GPU stress: compress -> bytes == oracle, decompress -> round trip, over random grid sizes, ratios (all 7 modes) and
code tables (max code length 13 / 17 / 128 / 224 bits).  Not a pytest (minutes); run by hand."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import control_gic_amd as cg
from oracle import cgic_oracle as orc
g = np.load(os.path.join(ROOT, "tests", "golden", "coders.npz"))
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60
cg._lib.call("cgic_set_decode_mode", {"auto": 0, "latency": 1, "throughput": 2}[sys.argv[3] if len(sys.argv) > 3 else "auto"])   # 3rd argument: decoder


class _Item:
    def __init__(self, v): self.v = v
    def item(self): return self.v


def mapping(freq, order):
    return {str(int(k)): _Item(float(freq[int(k)])) for k in order}


tables = {}
for name in ("zipf", "big", "ties", "zeros"):
    tables[name] = (mapping(g[name + "_freq"], g[name + "_order"]), orc.HuffmanTable(g[name + "_freq"]))
cbk = torch.from_numpy(rng.standard_normal((1024, 4)).astype(np.float32)).cuda()
codecs = {n: cg.GrainCodec(tables[n][0], cbk) for n in tables}
RATIOS = [(0.1, 0.8), (0.1, 0.4), (0.7, 0.3), (0.3, 0.7), (0.0, 0.4), (0.4, 0.0), (1.0, 0.0), (0.0, 1.0), (0.0, 0.0), (0.5, 0.5), (0.33, 0.33)]
t0 = time.time(); n = 0
while time.time() - t0 < budget:
    B = int(rng.integers(1, 6)); h = 4 * int(rng.integers(1, 50)); w = 4 * int(rng.integers(1, 50))
    name = str(rng.choice(list(tables), p=[0.5, 0.3, 0.1, 0.1]))
    if name in ("ties", "zeros") and h * w > 64 * 64: h, w = 32, 48          # 128/224-bit codes take the one-wave path
    c, m = RATIOS[int(rng.integers(0, len(RATIOS)))]
    e16 = torch.from_numpy((rng.random((B, h // 4, w // 4)) * 2.6).astype(np.float32)).cuda()
    e8 = torch.from_numpy((rng.random((B, h // 2, w // 2)) * 2.6).astype(np.float32)).cuda()
    ind = rng.integers(0, 1024, (B, h, w))
    mask, _, _, mode = cg.TripleGrainFixedEntropyRouter(c, m, per_image=True)(e16, e8)
    comp = codecs[name].compress(torch.from_numpy(ind).cuda(), mask, mode)
    host = comp.to_host()
    mks = [t.cpu().numpy() for t in mask]
    b = int(rng.integers(0, B))
    ref = orc.compress_image(ind[b], mks[0][b, 0], mks[1][b, 0], mks[2][b, 0], mode, tables[name][1])
    ok = host[b] == ref
    dind, dmask, zq, status = codecs[name].decompress(comp)
    exp = np.where(mks[2][:, 0] == 1, ind, 0)
    exp = exp + np.repeat(np.repeat(np.where(mks[1][:, 0] == 1, ind[:, ::2, ::2], 0), 2, 1), 2, 2)
    exp = exp + np.repeat(np.repeat(np.where(mks[0][:, 0] == 1, ind[:, ::4, ::4], 0), 4, 1), 4, 2)
    ok = ok and int(status.abs().max()) == 0 and np.array_equal(dind.cpu().numpy(), exp)
    ok = ok and all(torch.equal(a, b_) for a, b_ in zip(dmask, mask))
    if not ok:
        print("MISMATCH", dict(B=B, h=h, w=w, table=name, ratio=(c, m), mode=mode, image=b)); sys.exit(1)
    n += 1
print(f"{n} random cases: bytes == oracle and decode == merge of the encoded grids everywhere")

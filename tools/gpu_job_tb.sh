cd /root/repo
timeout 900 python -m pytest tests/test_highres_container.py -q -x -m gpu -k "tiled" 2>&1 | tail -3
timeout 600 python - <<'PY' 2>&1 | grep "ms_per_image\|MPixels\|error\|ok"
import json, torch, numpy as np, bench
import control_gic_amd as cg
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
cb = torch.from_numpy(rng.standard_normal((1024, 4)).astype(np.float32)).to(dev)
vq = cg.VectorQuantizer(1024, 4, beta=0.25).to(dev)
with torch.no_grad(): vq.embedding.weight.copy_(cb)
vq.usage_counter.copy_(torch.from_numpy(rng.integers(1, 1000, 1024).astype(np.float32)))
codec = cg.GrainCodec(vq.embedding_counter, vq.embedding.weight.detach())
print(json.dumps(bench.div2k_image(dev, cb, vq, codec), indent=1))
PY

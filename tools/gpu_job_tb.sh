#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import json, torch, numpy as np, bench
import control_gic_amd as cg
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
cb = torch.from_numpy(rng.standard_normal((1024, 4)).astype(np.float32)).to(dev)
vq = cg.VectorQuantizer(1024, 4, beta=0.25).to(dev)
with torch.no_grad(): vq.embedding.weight.copy_(cb)
vq.usage_counter.copy_(torch.from_numpy(rng.integers(1, 1000, 1024).astype(np.float32)))
codec = cg.GrainCodec(vq.embedding_counter, vq.embedding.weight.detach())
r = bench.div2k_image(dev, cb, vq, codec)
for k in ("batch_of_8", "batch_of_8_uint8_frames"):
    print(k, {a: b for a, b in r[k].items() if a != "note"})
PY

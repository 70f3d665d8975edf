#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
python - <<'PY'
import json, sys, torch
sys.argv=["bench.py"]
import bench, control_gic_amd as cg
dev=torch.device("cuda",0)
x,z,cb=bench.make_inputs(4,256,256,seed=1)
vq=bench.make_quantizer(dev,cb); codec=cg.GrainCodec(vq.embedding_counter, vq.embedding.weight)
print(json.dumps(bench.div2k_image(dev,cb,vq,codec), indent=1))
PY

"""GPU probe: the five launches of B images of SxS (default one 256x256 image), each timed alone (20 per hipGraph), and the whole chain
per graph: probe_b1.py [B [S]]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench, control_gic_amd as cg
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
x, z, cb = bench.make_inputs(B, S, S, 5)
vq = bench.make_quantizer(dev, cb)
codec = cg.GrainCodec(vq.embedding_counter, vq.embedding.weight)
hp = bench.HotPath(dev, x, z, cb, (0.1, 0.8), vq=vq, codec=codec)
hp.step(); torch.cuda.synchronize()
st = bench.stage_breakdown(hp)
for k, v in st.items(): print(f"  {k}: {v:.2f} us")
chain = bench.graph_kernel_time(hp.step, per_graph=10)
print(f"B={B} S={S}: whole step, 10 per graph: {chain:.2f} us per step; sum of the five launches alone: "
      f"{st['entropy_maps'] + st['vq+router_fused_launch'] + st['compress_streams+hist'] + st['decompress_streams']:.2f} us")

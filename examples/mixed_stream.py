#!/usr/bin/env python3
"""BASELINE config 5 in miniature: a mixed stream of Kodak-sized (768x512) and DIV2K-sized (2040x1356, tiled)
synthetic images, sharded round-robin over the ranks of one node, ONE RCCL all-reduce of the int64[1024] usage
histogram at the end and a 2-scalar reduction for the dataset-average bpp.

    python examples/mixed_stream.py                       # 1 GPU
    python examples/mixed_stream.py --gpus 8              # spawns the 8 ranks itself (one per GPU, RCCL)
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/mixed_stream.py --gpus 8

The conv encoder of the codec is out of scope: `encode()` below stands in for it with a cheap deterministic
latent (average-pooled image channels), then runs the real hot path: entropy maps -> [VQ + per-tile router]
-> stream coder; tiles of equal shape go through the kernels as one batch.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import control_gic_amd as cg                                                       # noqa: E402
from control_gic_amd import container, dist as cdist, highres                      # noqa: E402
from control_gic_amd.quantize import vq_forward_route                              # noqa: E402


def run(rank, world, expect_world=None):
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if torch.cuda.device_count() <= local:
        raise SystemExit(f"mixed_stream: rank {rank} needs GPU {local}, {torch.cuda.device_count()} visible")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if "RANK" in os.environ:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        joined = torch.ones(1, dtype=torch.int64, device=dev)
        torch.distributed.all_reduce(joined)
        if int(joined.item()) != world:
            raise SystemExit(f"mixed_stream: {int(joined.item())} ranks joined the communicator, expected {world}")

    sizes = [(512, 768)] * 6 + [(1356, 2040)] * 2 + [(512, 768)] * 6 + [(1356, 2040)] * 2      # (H, W) stream order
    vq = cg.VectorQuantizer(1024, 4, beta=0.25).to(dev).eval()
    vq.embedding.weight.data.copy_(torch.from_numpy(np.random.default_rng(12345).standard_normal((1024, 4), dtype=np.float32)))
    vq.usage_counter.copy_(torch.from_numpy(np.floor(2.0e6 / (1 + np.arange(1024)) ** 1.1).astype(np.float32)))
    codec = cg.GrainCodec(vq.embedding_counter, vq.embedding.weight)
    hist = torch.zeros(1024, dtype=torch.int64, device=dev)

    def encode(tiles):
        e8, e16 = cg.entropy_maps(tiles)
        z = torch.nn.functional.avg_pool2d(torch.cat([tiles, tiles[:, :1]], 1), 4) * 4 - 2     # stand-in latent [T,4,h,w]
        _, _, ind, mask, _, mode = vq_forward_route(z, vq.embedding.weight, 0.25, True, e16, e8, 0.1, 0.8,
                                                    per_image=True, want_zq=False, want_loss=False, pixels=tiles)
        cg._lib.call("cgic_index_histogram", ind.data_ptr(), ind.numel(), 1024, hist.data_ptr(), cg._lib.current_stream(dev))
        return ind, mask, mode

    bpps = []
    blobs = []
    mine = cdist.shard(len(sizes), rank, world)
    work = None
    for i in mine:
        H, W = sizes[i]
        x = torch.from_numpy(np.random.default_rng(100 + i).random((1, 3, H, W), dtype=np.float32)).to(dev)
        x = x[:, :, :H // 16 * 16, :W // 16 * 16] if H <= 768 and W <= 768 else x       # small images: centre-crop rule of inference.py:66-70
        tiled = highres.compress_tiled(x, encode, codec)
        entries = container.entries_from_tiled(tiled, image_id=i)
        blobs.append(container.pack(entries))
        bpps.append(tiled.bpp())                              # per image, the reference's accounting (unpadded pixels)
        if i == mine[-1]:
            # the histogram is complete once the LAST image is encoded: the path's only collective goes out now
            # (async_op: RCCL's own stream, ordered after the encode kernels) and runs under this image's decode side
            work = cdist.all_reduce_histogram(hist, async_op=True)
        # decode side on the same rank: every tile must come back (masks exactly; indices wherever the fine grain kept them)
        per_tile, _ = highres.decompress_tiled(tiled, codec)
        for idxs, _, (ind0, masks0, _) in tiled.groups:
            T = len(idxs)
            fine = masks0[2].reshape(T, -1).bool()
            got = torch.cat([per_tile[t][0].reshape(1, -1) for t in idxs])
            assert torch.equal(got[fine], ind0.reshape(T, -1)[fine])
            for g in range(3):
                assert torch.equal(torch.cat([per_tile[t][1][g] for t in idxs]).reshape(T, -1), masks0[g].reshape(T, -1))
    if work is not None:
        work.wait()                                        # (None: one rank, or no image on this rank)
    elif not mine:
        cdist.all_reduce_histogram(hist)                   # a rank without images still takes part in the collective
    bpp = cdist.average_bpp(bpps, device=dev)              # unweighted mean over images, like inference.py:168-171
    if rank == 0:
        total = int(hist.sum())
        want = sum(((h + 15) // 16 * 4) * ((w + 15) // 16 * 4) if (h > 768 or w > 768) else (h // 16 * 4) * (w // 16 * 4) for h, w in sizes)
        print(f"ranks {world}: {len(sizes)} images, average bpp {bpp:.5f}, histogram total {total} (latent vectors: {want}), "
              f"container bytes on rank 0: {sum(map(len, blobs))}")
        assert total == want
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks (one per GPU); default: WORLD_SIZE or 1")
    a = ap.parse_args()
    if "RANK" in os.environ:                                   # under torch.distributed.run
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if a.gpus is not None and a.gpus != world:
            raise SystemExit(f"mixed_stream: --gpus {a.gpus} but WORLD_SIZE={world}")
        run(int(os.environ["RANK"]), world)
    elif a.gpus in (None, 1):
        run(0, 1)
    else:
        if torch.cuda.device_count() < a.gpus:
            raise SystemExit(f"mixed_stream: --gpus {a.gpus} but only {torch.cuda.device_count()} GPUs are visible")
        cdist.spawn_ranks(run, a.gpus)


if __name__ == "__main__":
    main()

"""Multi-GPU support for the hot path: one process per GPU, `torch.distributed` (backend "nccl" = RCCL
over xGMI on ROCm; "gloo" in the CPU tests).

The path shards naturally -- every image (and every 768x768 tile) is independent end to end: router
thresholds, VQ and all five streams are per image, the Huffman table is static and replicated
(SURVEY.md section 8e).  So there is NO data-path collective.  The only exchanges are
  * one all-reduce(SUM) of the int64[n_e] usage histogram per stream of batches -- the correct
    multi-GPU version of the reference's per-rank usage counter (quantize.py:28,79-81, which DDP never
    synchronises because the counters are requires_grad=False), kept in int64 so it stays exact
    (the reference's fp32 counters stop at 2**24);
  * a 2-element reduction (sum of per-image bpp, image count) for the dataset-average bpp that inference.py:168-171 prints.
An 8 KB all-reduce is latency-bound on the 7-link xGMI mesh (tens of microseconds); it is issued once per
stream of batches, never per image.
"""
import os

import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard(n_items, rank=None, world_size=None):
    """indices of the images / tiles this rank owns: round-robin, so that mixed-size streams
    (config 5: Kodak + DIV2K) balance without knowing the sizes"""
    r, w = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    return list(range(rank, n_items, world_size))


def all_reduce_histogram(hist, async_op=False):
    """in-place SUM over ranks of an int64 [n_e] histogram; returns the work handle if async_op"""
    if hist.dtype != torch.int64:
        raise TypeError("the usage histogram is exchanged as int64 (exact); got " + str(hist.dtype))
    _, w = world()
    if w == 1:
        return None
    return dist.all_reduce(hist, op=dist.ReduceOp.SUM, async_op=async_op)


def fold_histogram_into(quantizer, hist):
    """add a (globally reduced) histogram to a VectorQuantizer's fp32 usage counters, the tensor
    HuffmanCoding(model.quantize.embedding_counter) is built from (inference.py:137-139)"""
    with torch.no_grad():
        quantizer.usage_counter += hist.to(quantizer.usage_counter)


def average_bpp(per_image_bpp, device=None):
    """dataset-average bits per pixel over all ranks as the reference reports it: the UNWEIGHTED mean of the per-image
    bpp values, `bpp_sum / len(dataset)` (inference.py:168-171, inference_high_resolution.py:259-262) -- images of
    different sizes count equally.  per_image_bpp: this rank's list of per-image bpp."""
    vals = [float(v) for v in per_image_bpp]
    t = torch.tensor([sum(vals), float(len(vals))], dtype=torch.float64, device=device)
    _, w = world()
    if w > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t[0] / t[1]) if float(t[1]) > 0 else float("nan")


def pixel_weighted_bpp(local_bits, local_pixels, device=None):
    """total bits / total pixels over all ranks (NOT the reference's dataset average when image sizes differ;
    it is the bpp of the whole stream seen as one image)"""
    t = torch.tensor([float(local_bits), float(local_pixels)], dtype=torch.float64, device=device)
    _, w = world()
    if w > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t[0] / t[1])


def spawn_ranks(fn, world_size, args=(), port=None):
    """run fn(rank, world_size, *args) in `world_size` fresh processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set,
    for scripts started plainly (`python script.py --gpus N`) instead of under torch.distributed.run; raises if any
    rank fails"""
    import socket
    import torch.multiprocessing as mp
    if port is None:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    mp.spawn(_spawned, args=(fn, world_size, port, args), nprocs=world_size, join=True)


def _spawned(rank, fn, world_size, port, args):
    os.environ["RANK"] = os.environ["LOCAL_RANK"] = str(rank)
    os.environ["WORLD_SIZE"] = str(world_size)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    fn(rank, world_size, *args)

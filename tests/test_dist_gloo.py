"""CPU, world_size 2, gloo: the N>1 path -- sharding covers every image exactly once, the int64 usage
histogram all-reduce is exact, and the average-bpp reduction matches the serial value."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_images, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from control_gic_amd import dist as cdist
    mine = cdist.shard(n_images)
    # deterministic synthetic "indices" per image; every rank can regenerate any image
    hist = torch.zeros(1024, dtype=torch.int64)
    bits = pixels = 0
    for i in mine:
        rng = np.random.default_rng(1000 + i)
        idx = rng.integers(0, 1024, 4096)
        idx[:50] = 7                                  # a hot bin
        hist += torch.from_numpy(np.bincount(idx, minlength=1024))
        bits += int(rng.integers(10_000, 20_000))
        pixels += 65536
    big = torch.zeros(1024, dtype=torch.int64)
    big[3] = 2 ** 40 + rank                           # beyond fp32's exact range: must stay exact
    cdist.all_reduce_histogram(hist)
    cdist.all_reduce_histogram(big)
    bpp = cdist.average_bpp(bits, pixels)
    torch.save({"mine": mine, "hist": hist, "big": big, "bpp": bpp}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_histogram_and_bpp_world2(tmp_path):
    world, n_images = 2, 13
    mp.spawn(_worker, args=(world, _free_port(), n_images, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(tmp_path / f"r{r}.pt") for r in range(world)]
    owned = sorted(i for r in res for i in r["mine"])
    assert owned == list(range(n_images))                       # every image exactly once
    ref = np.zeros(1024, np.int64)
    bits = pixels = 0
    for i in range(n_images):
        rng = np.random.default_rng(1000 + i)
        idx = rng.integers(0, 1024, 4096)
        idx[:50] = 7
        ref += np.bincount(idx, minlength=1024)
        bits += int(rng.integers(10_000, 20_000))
        pixels += 65536
    for r in res:
        assert np.array_equal(r["hist"].numpy(), ref)           # identical, exact, on every rank
        assert int(r["big"][3]) == 2 * 2 ** 40 + 1
        assert abs(r["bpp"] - bits / pixels) < 1e-15


def test_single_process_is_a_noop():
    from control_gic_amd import dist as cdist
    assert cdist.world() == (0, 1)
    assert cdist.shard(5) == [0, 1, 2, 3, 4]
    h = torch.arange(1024)
    assert cdist.all_reduce_histogram(h) is None and int(h[5]) == 5
    assert cdist.average_bpp(100, 50) == 2.0

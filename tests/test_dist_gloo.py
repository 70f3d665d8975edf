"""CPU, world_size 2, gloo: the N>1 path -- sharding covers every image exactly once, the int64 usage
histogram all-reduce is exact, and the average-bpp reduction matches the serial value."""
import os

import pytest
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
    bpps = []
    for i in mine:
        rng = np.random.default_rng(1000 + i)
        idx = rng.integers(0, 1024, 4096)
        idx[:50] = 7                                  # a hot bin
        hist += torch.from_numpy(np.bincount(idx, minlength=1024))
        b, px = int(rng.integers(10_000, 20_000)), 65536 * (1 + i % 3)     # images of different sizes
        bits += b
        pixels += px
        bpps.append(b / px)
    big = torch.zeros(1024, dtype=torch.int64)
    big[3] = 2 ** 40 + rank                           # beyond fp32's exact range: must stay exact
    cdist.all_reduce_histogram(hist)
    cdist.all_reduce_histogram(big)
    bpp = cdist.average_bpp(bpps)
    wbpp = cdist.pixel_weighted_bpp(bits, pixels)
    torch.save({"mine": mine, "hist": hist, "big": big, "bpp": bpp, "wbpp": wbpp}, os.path.join(out_dir, f"r{rank}.pt"))
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
    bpps = []
    for i in range(n_images):
        rng = np.random.default_rng(1000 + i)
        idx = rng.integers(0, 1024, 4096)
        idx[:50] = 7
        ref += np.bincount(idx, minlength=1024)
        b, px = int(rng.integers(10_000, 20_000)), 65536 * (1 + i % 3)
        bits += b
        pixels += px
        bpps.append(b / px)
    for r in res:
        assert np.array_equal(r["hist"].numpy(), ref)           # identical, exact, on every rank
        assert int(r["big"][3]) == 2 * 2 ** 40 + 1
        # the reference's dataset average: unweighted mean of per-image bpp (inference.py:168-171) ...
        assert abs(r["bpp"] - sum(bpps) / len(bpps)) < 1e-12
        # ... which is NOT total bits / total pixels when sizes differ
        assert abs(r["wbpp"] - bits / pixels) < 1e-15 and abs(r["bpp"] - r["wbpp"]) > 1e-3


def test_single_process_is_a_noop():
    from control_gic_amd import dist as cdist
    assert cdist.world() == (0, 1)
    assert cdist.shard(5) == [0, 1, 2, 3, 4]
    h = torch.arange(1024)
    assert cdist.all_reduce_histogram(h) is None and int(h[5]) == 5
    assert cdist.average_bpp([2.0, 4.0]) == 3.0 and cdist.pixel_weighted_bpp(100, 50) == 2.0


def _spawn_target(rank, world, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.ones(1, dtype=torch.int64)
    dist.all_reduce(t)
    with open(os.path.join(out_dir, f"spawned{rank}.txt"), "w") as f:
        f.write(f"{os.environ['RANK']} {os.environ['LOCAL_RANK']} {os.environ['WORLD_SIZE']} {int(t)}")
    dist.destroy_process_group()


def test_spawn_ranks_helper(tmp_path):
    """dist.spawn_ranks: what `python examples/mixed_stream.py --gpus N` uses when it is not under torch.distributed.run"""
    from control_gic_amd import dist as cdist
    cdist.spawn_ranks(_spawn_target, 2, args=(str(tmp_path),))
    for r in range(2):
        assert open(tmp_path / f"spawned{r}.txt").read() == f"{r} {r} 2 2"


def _bench(args, env_extra=None, timeout=300):
    """bench.py in a process GROUP of its own, output through files: on a timeout the whole group is killed -- ranks that bench.py
    spawned included -- and nothing waits for a pipe a stray grandchild still holds (subprocess.run(capture_output=True, timeout=...)
    kills the child and then blocks on exactly that)"""
    import signal
    import subprocess
    import sys
    import tempfile
    import types
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    with tempfile.TemporaryFile("w+") as out, tempfile.TemporaryFile("w+") as err:
        p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py")] + args, stdout=out, stderr=err, text=True, env=env,
                             start_new_session=True)
        try:
            rc = p.wait(timeout=timeout)
        except subprocess.TimeoutExpired:
            os.killpg(p.pid, signal.SIGKILL)
            p.wait()
            rc = -9
        try:
            os.killpg(p.pid, signal.SIGKILL)          # (whatever the group still holds: a rank stuck in its teardown)
        except ProcessLookupError:
            pass
        out.seek(0); err.seek(0)
        return types.SimpleNamespace(returncode=rc, stdout=out.read(), stderr=err.read())


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` without torch.distributed.run starts 2 ranks itself (here: the CPU stub of the hot
    path over gloo), rank 0 prints ONE JSON line with n_gpus == rccl_ranks == 2 and the whole-job value"""
    import json
    r = _bench(["--gpus", "2", "--stub", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["steps"] == 3 and d["scaling"] == "weak"
    assert abs(d["value"] - 2 * 3 * 64 * 256 * 256 / (d["ms_per_step"] * 3 * 1e-3) / 1e6) / d["value"] < 1e-3
    # per-rank rates and the collective's own time are on the line (the driver computes scaling efficiency from `value`)
    assert len(d["per_rank_MPixels/s"]) == 2 and min(d["per_rank_MPixels/s"]) * 2 >= d["value"] * 0.999 and d["histogram_allreduce_us"] > 0
    # the timed region holds exactly ONE collective (the histogram all-reduce, which is also its closing barrier): counted by
    # intercepting torch.distributed while the region runs; every rank's own K steps by its own clock are on the line as well
    assert d["timed_collectives"] == 1 and len(d["per_rank_MPixels/s_device_events"]) == 2
    # N = 1 takes the same path: a one-rank communicator, `rccl_ranks` read back from it
    one = json.loads(_bench(["--gpus", "1", "--stub", "--steps", "3", "--warmup", "1"]).stdout.strip())
    assert one["n_gpus"] == 1 and one["rccl_ranks"] == 1 and len(one["per_rank_MPixels/s"]) == 1 and one["histogram_allreduce_us"] is not None
    assert one["timed_collectives"] == 0                         # one rank: nothing to exchange inside the region
    solo = json.loads(_bench(["--gpus", "1", "--stub", "--steps", "3", "--warmup", "1", "--no-dist"]).stdout.strip())
    assert solo["rccl_ranks"] is None and solo["histogram_allreduce_us"] is None


def test_bench_mixed_workload_shards_the_stream():
    """`bench.py --workload mixed` (BASELINE config 5: 24 Kodak-sized + N DIV2K-sized images, round-robin over the ranks, ONE
    histogram all-reduce issued async under the last decode side): launcher, sharding and reduction logic over gloo with the CPU
    stand-in for the kernels -- strong scaling (the stream is fixed), exact histogram total, per-rank rates"""
    import json
    import bench
    sizes = bench.mixed_sizes(8)
    assert len(sizes) == 32 and sizes.count(bench.MIXED_DIV2K) == 8
    shares = [bench.mixed_share(sizes, r, 2) for r in range(2)]
    assert sum(s[0] for s in shares) == 24 and sum(s[1] for s in shares) == 8 and shares[0][:2] == shares[1][:2] == (12, 4)
    for world in (3, 4, 8):
        sh = [bench.mixed_share(sizes, r, world) for r in range(world)]
        assert sum(v[0] for v in sh) == 24 and sum(v[1] for v in sh) == 8 and max(v[2] for v in sh) <= 1.35 * min(v[2] for v in sh)
    assert bench.mixed_share(sizes, 0, 1)[3] == 24 * 128 * 192 + 8 * 340 * 512          # latent vectors (DIV2K padded to 1360 x 2048)
    for gpus in (1, 2):
        r = _bench(["--gpus", str(gpus), "--stub", "--workload", "mixed", "--steps", "3", "--warmup", "1"])
        assert r.returncode == 0, r.stdout + r.stderr
        lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1, r.stdout
        d = json.loads(lines[0])
        assert d["n_gpus"] == gpus and d["rccl_ranks"] == gpus and d["scaling"] == "strong" and "mixed stream" in d["config"]["workload"]
        assert d["timed_collectives"] == (1 if gpus > 1 else 0)            # the async histogram all-reduce and nothing else
        pix = 24 * 512 * 768 + 8 * 1356 * 2040
        assert abs(d["value"] - 3 * pix / (d["ms_per_step"] * 3 * 1e-3) / 1e6) / d["value"] < 1e-3 and len(d["per_rank_MPixels/s"]) == gpus


def test_bench_refuses_a_mismatched_world():
    """under a launcher whose WORLD_SIZE differs from --gpus, or without enough GPUs, bench.py exits non-zero instead of
    silently benchmarking fewer GPUs than it reports"""
    r = _bench(["--gpus", "2", "--stub", "--steps", "2"], {"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr + r.stdout
    import torch as _t
    if not _t.cuda.is_available():
        r = _bench(["--gpus", "2", "--steps", "2"])
        assert r.returncode != 0 and "GPUs are visible" in r.stderr + r.stdout


def _counter_worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import control_gic_amd as cg
    vq = cg.VectorQuantizer(1024, 4, beta=0.25).train()          # CPU module: no kernel is launched below
    assert vq.sync_usage_counter is False                          # the default is the reference's: no collective inside forward()
    vq.sync_usage_counter = True                                   # opt in: every step's histogram summed over the ranks
    rng = np.random.default_rng(50 + rank)
    total = np.zeros(1024, np.int64)
    for step in range(3):                                          # what forward() does after the kernel filled usage_hist
        h = np.bincount(rng.integers(0, 1024, 4096), minlength=1024)
        vq.usage_hist += torch.from_numpy(h)
        vq.fold_usage_hist()
        total += h
    local = cg.VectorQuantizer(1024, 4, beta=0.25).train()        # the default = the reference: every rank counts its own shard
    local.usage_hist += torch.from_numpy(total)
    local.fold_usage_hist()
    late = cg.VectorQuantizer(1024, 4, beta=0.25).train()         # per-rank counting, ONE reduction at "checkpoint time"
    late.usage_hist += torch.from_numpy(total)
    late.fold_usage_hist()
    late.sync_usage_counter_now()
    # a checkpoint of the whole job's counts, loaded on every rank, module moved afterwards, synced twice: nothing to add
    ckpt = cg.VectorQuantizer(1024, 4, beta=0.25)
    ckpt.usage_counter.copy_(torch.arange(1024, dtype=torch.float32) * 3)
    loaded = cg.VectorQuantizer(1024, 4, beta=0.25).train()
    loaded.load_state_dict(ckpt.state_dict())
    loaded = loaded.to(torch.device("cpu"))
    loaded.sync_usage_counter_now()
    loaded.sync_usage_counter_now()
    assert torch.equal(loaded.usage_counter, ckpt.usage_counter), "checkpointed counts were multiplied by the world size"
    loaded.usage_hist += torch.from_numpy(total)                   # then each rank counts its shard: only that delta is summed
    loaded.fold_usage_hist()
    loaded.sync_usage_counter_now()
    torch.save({"loaded": loaded.usage_counter.clone(), "ckpt": ckpt.usage_counter.clone(), "synced": vq.usage_counter.clone(), "local": local.usage_counter.clone(), "mine": torch.from_numpy(total), "late": late.usage_counter.clone(),
                "hist_left": int(vq.usage_hist.abs().sum()), "key3": float(vq.embedding_counter["3"].item())},
               os.path.join(out_dir, f"c{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_training_usage_counter_is_reduced_over_ranks(tmp_path):
    """VectorQuantize2 in training mode under a process group: every rank's embedding_counter counts the WHOLE batch
    (sum over ranks, exact), identically on all ranks, with sync_usage_counter=True (a collective per step) or with one
    sync_usage_counter_now() at the end; the default keeps the reference's per-rank counts and puts no collective in forward()"""
    world = 2
    mp.spawn(_counter_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(tmp_path / f"c{r}.pt") for r in range(world)]
    want = (res[0]["mine"] + res[1]["mine"]).float()
    for r in res:
        assert torch.equal(r["synced"], want) and r["hist_left"] == 0 and r["key3"] == float(want[3])
        assert torch.equal(r["local"], r["mine"].float())
        assert torch.equal(r["late"], want)                      # sync_usage_counter_now(): the same table from one collective
        assert torch.equal(r["loaded"], r["ckpt"] + want)        # checkpoint base once + the ranks' deltas
    assert not torch.equal(res[0]["local"], res[1]["local"])


def test_bench_dry_nccl_spawns_the_single_rank_like_n_ranks():
    """`bench.py --gpus 1 --dry-nccl`: the one rank runs in a child started by torch.multiprocessing.spawn with RANK / WORLD_SIZE /
    MASTER_* set by the parent -- the code path of `--gpus N` for N > 1 -- instead of in the calling process (here over gloo with
    the CPU stand-in for the kernels; the GPU form of this test is below)"""
    import json
    r = _bench(["--gpus", "1", "--stub", "--steps", "3", "--warmup", "1", "--dry-nccl"])
    assert r.returncode == 0, r.stdout + r.stderr
    d = json.loads(r.stdout.strip())
    assert d["n_gpus"] == 1 and d["rccl_ranks"] == 1 and d["spawned_ranks"] is True and d["timed_collectives"] == 0


@pytest.mark.gpu
def test_bench_dry_nccl_on_the_gpu():
    """mp.spawn + init_process_group("nccl", device_id=...) + the RCCL all-reduce in a child process, exactly as
    the ranks of an N > 1 run are started, exercised at N = 1 on whatever box runs the GPU tests; the line names the rank's PCI bus
    id and the RCCL version"""
    import json
    r = _bench(["--gpus", "1", "--dry-nccl", "--steps", "6", "--warmup", "2", "--no-extra", "--no-cpu-baseline"], timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["rccl_ranks"] == 1 and d["spawned_ranks"] is True and d["bpp_match"] is True
    assert d["rank_pci_bus_ids"][0] and d["rccl_version"] and d["histogram_allreduce_us"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["batch", "mixed"])
def test_bench_two_gpus_for_real(workload):
    """a NON-stub `--gpus 2` run of both workloads (RCCL over xGMI, one process per GPU) wherever two GPUs are visible; skipped on
    the one-GPU boxes of the pool"""
    import json
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible")
    r = _bench(["--gpus", "2", "--workload", workload, "--steps", "6", "--warmup", "2", "--no-extra", "--no-cpu-baseline"], timeout=400)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["timed_collectives"] == 1
    assert len(set(d["rank_pci_bus_ids"])) == 2 and len(d["per_rank_MPixels/s"]) == 2

"""The N>1 path on CPU: world-size-2 gloo run of the sharding logic (batch ownership, host noise
replay in the reference's order, the single all-gather) with a stand-in per-batch function."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from diffusion_nlc_amd import shard

SHAPE = (2, 3, 4, 4)


def _fake_sampler(z):
    return z * 2.0 + 1.0          # any per-sample-independent map


def _worker(rank, world, port, n_batches, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, w, _ = shard.init_from_env("gloo")
    out = shard.sample_sharded(_fake_sampler, SHAPE, n_batches, seed=1234, world=w, rank=r, device="cpu")
    if r == 0:
        q.put(out.numpy())          # by value: torch tensors travel by fd and the producer may exit first
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("n_batches", [4, 5])
def test_two_ranks_equal_single_process(n_batches):
    single = shard.sample_sharded(_fake_sampler, SHAPE, n_batches, seed=1234, world=1, rank=0, device="cpu")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_batches, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = torch.from_numpy(q.get(timeout=120))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got.shape == (n_batches,) + SHAPE
    assert torch.equal(got, single)


def test_ownership_and_noise_order():
    assert shard.owned_batches(5, 2, 0) == [0, 2, 4] and shard.owned_batches(5, 2, 1) == [1, 3]
    g = torch.Generator().manual_seed(7)
    ref = [torch.randn(SHAPE, generator=g) for _ in range(4)]
    for r in range(2):
        mine = shard.draw_initial_noise(SHAPE, 4, 7, 2, r)
        assert all(torch.equal(a, ref[j]) for a, j in zip(mine, shard.owned_batches(4, 2, r)))


def test_bench_self_launches_its_ranks_under_gloo():
    """`python bench.py --gpus 2` with no launcher around it: the parent starts two rank processes itself (before it
    touches any device), they rendezvous over gloo on 127.0.0.1, shard the batches, all-gather, and rank 0's JSON line
    comes back through the parent with n_gpus = 2.  --dry-run replaces the sampling (which needs the HIP path) by a
    stand-in so that the launcher / sharding / gather plumbing is what runs here."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["NLC_BENCH_FORCE_CPU"] = "1"
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run", "--tiny"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["config"]["global_batch"] == 32 and "DRY RUN" in line["metric"]
    # a rank that dies must fail the whole command
    bad = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0            # no GPU here and no --dry-run: every rank refuses

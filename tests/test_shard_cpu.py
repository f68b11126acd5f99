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
    # diagnosability of the first multi-GPU run: the world size the process group really had, and every rank's phase times
    assert line["ranks_seen"] == 2
    pr = line["per_rank_seconds"]
    assert set(pr) == {"timed_region", "sampling", "all_gather", "final_barrier", "setup_untimed", "model_build_untimed"}
    assert all(len(v) == 2 for v in pr.values())
    for r_ in range(2):
        assert abs(pr["sampling"][r_] + pr["all_gather"][r_] + pr["final_barrier"][r_] - pr["timed_region"][r_]) < 5e-3
    assert abs(max(pr["timed_region"]) * 1e3 / 2 - line["ms_per_step"]) < 1.0          # value = max over ranks
    # a rank that dies must fail the whole command
    bad = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0            # no GPU here and no --dry-run: every rank refuses


# ---- the reference's own entry points, sharded: image_sample.evaluate_unconstraint and EDMImageExperiment.evaluate_edm -----------------
# The HIP path has no CPU fallback, so a stand-in experiment supplies the per-batch sampler; everything else - batch ownership, replay
# of the single host generator for batches another rank owns, the once-only skip-if-exists decision, the all-gather, rank 0 writing the
# files - is the CLI's own code.

class _FakeExperiment:
    """Quacks like ImageExperiment for evaluate_unconstraint: draws the initial state from `gen`, then `noisy_steps` per-step draws
    from the global generator (what stochastic samplers do, src/experiments.py:268 and the schedulers' randn calls)."""
    device = "cpu"
    fid_fn = None

    def __init__(self, noisy_steps):
        self.batch_size, self.data_shape, self.seed, self.noisy_steps = 2, (3, 4, 4), 99, noisy_steps

    def new_gen(self):
        return torch.manual_seed(self.seed)

    def host_draws_per_batch(self, new_eta=None):
        return 1 + self.noisy_steps

    def denoise_loop(self, shape, gen=None, return_on_device=False, **kw):
        x = torch.randn(shape, generator=gen)
        for _ in range(self.noisy_steps):
            x = 0.7 * x + 0.3 * torch.randn(shape)
        return torch.tanh(x), []


def _eval_worker(rank, world, port, noisy_steps, images_dir, n_samples):
    import image_sample
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
        shard.init_from_env("gloo")
    image_sample.save_png = lambda img, p: torch.save(img.clone(), p)          # exact values instead of 8-bit PNGs
    image_sample.evaluate_unconstraint(_FakeExperiment(noisy_steps), n_samples, images_dir, save_images=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _run_eval(world, noisy_steps, images_dir, n_samples=9):
    if world == 1:
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            os.environ.pop(k, None)
        _eval_worker(0, 1, 0, noisy_steps, images_dir, n_samples)
        return
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_eval_worker, args=(r, world, port, noisy_steps, images_dir, n_samples)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0


@pytest.mark.parametrize("noisy_steps", [0, 3], ids=["deterministic-sampler", "stochastic-sampler"])
def test_evaluate_unconstraint_two_ranks_equal_single_process(tmp_path, noisy_steps):
    """image_sample.py's own evaluate function under WORLD_SIZE=2 (gloo): every file equals the single-process run's, including
    when one batch's files already exist (the reference skips it WITHOUT advancing the generator, image_sample.py:533-541)."""
    one, two = tmp_path / "one", tmp_path / "two"
    for d in (one, two):
        d.mkdir()
        for j in range(2):                                          # batch 1 of 5 is already there
            torch.save(torch.full((3, 4, 4), -7.0), d / f"00-00001-{j:03}.png")
    _run_eval(1, noisy_steps, str(one))
    _run_eval(2, noisy_steps, str(two))
    names = sorted(p.name for p in one.iterdir())
    assert names == sorted(p.name for p in two.iterdir()) and len(names) == 10            # ceil(9 / 2) = 5 full batches
    for n in names:
        a, b = torch.load(one / n), torch.load(two / n)
        assert torch.equal(a, b), n
    assert float(torch.load(two / "00-00001-000.png").mean()) == -7.0                    # the pre-existing batch was not touched
    # and the run really depends on the generator replay: batch 4 differs from batch 0
    assert not torch.equal(torch.load(two / "00-00004-000.png"), torch.load(two / "00-00000-000.png"))


def _edm_worker(rank, world, port, q, images_dir):
    import types
    from diffusion_nlc_amd.experiments import EDMImageExperiment
    from diffusion_nlc_amd import edm_experiment
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
        shard.init_from_env("gloo")
    edm_experiment.save_image = lambda img, path: torch.save(img.clone(), path)          # "PNG" = the exact tensor
    fake = types.SimpleNamespace(batch_size=2, data_shape=(3, 4, 4), device="cpu", fid_fn=None,
                                 edm_sampler=lambda shape, gen=None, **kw: gen.randn(shape, dtype=torch.float64) * 0.4)
    log, samples = EDMImageExperiment.evaluate_edm(fake, 10, images_dir, return_samples=True)
    assert set(log) == {"fid"}
    if rank == 0:
        q.put(samples.numpy())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_evaluate_edm_two_ranks_equal_single_process(tmp_path):
    """EDMImageExperiment.evaluate_edm under WORLD_SIZE=2: per-sample generators (seed = global sample index), to-do batches
    dealt round-robin, one all-gather, rank 0 writes every file under the single-process names; a batch whose files exist is
    skipped by both (src/experiments.py:923-961)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    one, two = tmp_path / "one", tmp_path / "two"
    for d in (one, two):
        d.mkdir()
        for j in range(2):                                          # batch 3 of 5 is already there
            torch.save(torch.full((3, 4, 4), -7.0), d / f"00-00003-{j:03}.png")
    _edm_worker(0, 1, 0, q, str(one))
    single = q.get(timeout=60)
    port = _free_port()
    procs = [ctx.Process(target=_edm_worker, args=(r, 2, port, q, str(two))) for r in range(2)]
    for p in procs:
        p.start()
    both = q.get(timeout=180)
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    assert both.shape == (8, 3, 4, 4) and (both == single).all()                        # 4 sampled batches, in batch order
    names = sorted(p.name for p in one.iterdir())
    assert names == sorted(p.name for p in two.iterdir()) and len(names) == 10
    for n in names:
        assert torch.equal(torch.load(one / n), torch.load(two / n)), n
    assert float(torch.load(two / "00-00003-001.png").mean()) == -7.0
    # sample j of batch i was drawn from the generator seeded with its GLOBAL index: batch 4 = seeds 8, 9
    g = torch.Generator().manual_seed(9)
    want = (torch.randn((3, 4, 4), generator=g, dtype=torch.float64) * 0.4).add(1).div(2).clamp(0, 1)
    assert torch.equal(torch.load(two / "00-00004-001.png"), want)


# ---- the exchange half of the sigma-net training: DDP's construction-time parameter broadcast and its bucketed
#      gradient averaging, as ImageExperiment.sync_sigma_parameters / average_sigma_gradients -------------------------------------------
def _ddp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    shard.init_from_env("gloo")
    from diffusion_nlc_amd.experiments import ImageExperiment
    g = torch.Generator().manual_seed(100 + rank)                     # every rank starts from DIFFERENT values
    shapes = [(128, 1024, 3, 3), (128,), (7, 5), (1,), (300, 300)]
    params = [torch.randn(s, generator=g) for s in shapes]
    grads = [torch.randn(s, generator=g) * (rank + 1) for s in shapes]
    ImageExperiment.sync_sigma_parameters(params, src=0, bucket_mb=1)           # 1 MB buckets: the 4.7 MB tensor is a bucket of its own
    ImageExperiment.average_sigma_gradients(grads, bucket_mb=1)
    q.put((rank, [p.numpy() for p in params], [x.numpy() for x in grads]))
    dist.barrier()
    dist.destroy_process_group()


def test_sigma_training_exchange_two_ranks():
    """After sync_sigma_parameters every rank holds rank 0's parameters; after average_sigma_gradients every rank holds the mean of
    the two ranks' gradients (src/experiments.py:645-652: DistributedDataParallel(bucket_cap_mb=128) on the sigma net)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict()
    for _ in range(2):
        r, params, grads = q.get(timeout=180)
        got[r] = (params, grads)
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    shapes = [(128, 1024, 3, 3), (128,), (7, 5), (1,), (300, 300)]
    g0, g1 = torch.Generator().manual_seed(100), torch.Generator().manual_seed(101)
    p0 = [torch.randn(s, generator=g0) for s in shapes]
    gr0 = [torch.randn(s, generator=g0) for s in shapes]
    _ = [torch.randn(s, generator=g1) for s in shapes]
    gr1 = [torch.randn(s, generator=g1) * 2 for s in shapes]
    for r in range(2):
        for a, b in zip(got[r][0], p0):
            assert torch.equal(torch.from_numpy(a), b)
        for a, x, y in zip(got[r][1], gr0, gr1):
            assert torch.allclose(torch.from_numpy(a), (x + y) / 2, rtol=0, atol=1e-6)

"""RCCL on the hardware at hand.  The sharded path (bench.py --gpus N, both CLIs under torchrun) is covered by world-size-2 gloo tests
on CPU; no multi-GPU box is available to this suite, so what can be checked on ONE MI355X is that the `nccl` (= RCCL) backend of this
image comes up with the environment the launch lines use and runs the very collectives that path issues - all_gather_into_tensor of the
samples, the all_reduce / all_gather of the timing vector, a barrier, a broadcast - on device tensors, in a child process (a process
group is process-wide state).  One rank: no xGMI traffic, but library, bootstrap, stream semantics and the dmabuf-IPC setting are real."""
import os
import subprocess
import sys
import textwrap
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu

CHILD = textwrap.dedent('''
    import os, sys, torch, torch.distributed as dist
    sys.path.insert(0, os.environ["NLC_ROOT"])
    from diffusion_nlc_amd import shard
    os.environ.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29517")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
    assert dist.get_backend() == "nccl" and shard.world_rank() == (1, 0)
    x = torch.arange(2 * 3 * 8 * 8, device=dev, dtype=torch.float32).view(2, 3, 8, 8)
    out = torch.empty_like(x)
    dist.all_gather_into_tensor(out.view(-1), x.view(-1))
    assert torch.equal(out, x)
    t = torch.tensor([1.5, 2.5], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.tolist() == [1.5, 2.5]
    b = torch.full((1 << 20,), 3.0, device=dev)
    dist.broadcast(b, src=0)
    dist.barrier()
    torch.cuda.synchronize()
    assert float(b.sum()) == 3.0 * (1 << 20)
    dist.destroy_process_group()
    print("RCCL_OK", torch.version.hip)
''')


def test_rccl_backend_comes_up_and_runs_the_sharded_paths_collectives():
    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, NLC_ROOT=str(root), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])

"""-m gpu, needs >= 2 MI355X in one box (skipped otherwise): the first multi-GPU machine this suite meets exercises RCCL.

SURVEY 8(e): per-frame data parallel, one process per GPU, no data-path collective at inference, ONE flat-bucket gradient
all-reduce per training step.  The CPU suite covers the same plumbing over gloo (tests/test_distributed_gloo.py,
tests/test_launch_ddp_gloo.py, tests/test_bench_launcher.py); nothing here extrapolates a scaling curve -- the driver measures it."""
import json
import os
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu

needs2 = pytest.mark.skipif(not torch.cuda.is_available() or torch.cuda.device_count() < 2, reason="needs two GPUs in one box")


def _bench(*extra, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *extra], capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                  # rank 0 prints ONE line
    return json.loads(lines[0])


@needs2
def test_bench_two_ranks_inference_is_weak_scaling_without_a_collective():
    d = _bench("--gpus", "2", "--steps", "5", "--warmup", "2", "--no-kernel-table", "--no-extra-lines", "--no-other-configs")
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["parallelism"] == "frame-dp2"
    one = _bench("--gpus", "1", "--steps", "5", "--warmup", "2", "--no-kernel-table", "--no-extra-lines", "--no-other-configs",
                 "--no-cpu-baseline")
    assert d["value"] > 1.2 * one["value"], (d["value"], one["value"])      # two ranks do more than one (no claim about how much)


@needs2
def test_bench_two_ranks_training_all_reduces_one_bucket_over_rccl():
    d = _bench("--gpus", "2", "--workload", "car.fhd.train", "--dtype", "bf16", "--steps", "5", "--warmup", "2")
    c = d["config"]
    assert d["n_gpus"] == 2 and c["parallelism"] == "ddp2" and c["allreduce_us"] is not None and c["allreduce_us"] > 0
    assert c["gradient_bucket_bytes"] > 5_000_000
    assert all(v == v for v in d["loss_last_step"].values())               # finite losses on rank 0


@needs2
def test_grad_bucket_all_reduce_over_rccl_matches_the_mean_of_the_ranks(tmp_path):
    """distributed.GradBucket over the nccl (= RCCL) backend, two processes, one GPU each: the averaged bucket equals the mean of
    the two ranks' gradients, including a parameter that has a gradient on one rank only."""
    code = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(sys.argv[1], "second.pytorch_amd"))
from second_amd import distributed as D
rank, local, world = D.init_from_env("nccl")
torch.cuda.set_device(local)
torch.manual_seed(0)
net = torch.nn.Sequential(torch.nn.Linear(8, 4), torch.nn.Linear(4, 2)).cuda()
D.broadcast_parameters(net, 0)
bucket = D.GradBucket(net)
x = torch.full((3, 8), float(rank + 1), device="cuda")
y = net[0](x).sum() if rank == 0 else net(x).sum()          # rank 0 leaves the second layer without a gradient
y.backward()
mine = [None if p.grad is None else p.grad.clone() for p in net.parameters()]
bucket.allreduce(average=True)
got = [p.grad.clone() for p in net.parameters()]
gathered = [None, None]
dist.all_gather_object(gathered, [None if g is None else g.cpu() for g in mine])
if rank == 0:
    for i, g in enumerate(got):
        parts = [torch.zeros_like(g.cpu()) if gathered[r][i] is None else gathered[r][i] for r in range(2)]
        assert torch.allclose(g.cpu(), (parts[0] + parts[1]) / 2, rtol=1e-6, atol=1e-7), i
    print("RCCL_OK")
dist.destroy_process_group()
'''
    path = tmp_path / "rccl_bucket.py"
    path.write_text(code)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29617", str(path), ROOT], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])

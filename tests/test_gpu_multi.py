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


@needs2
@pytest.mark.parametrize("in_graph", ["1", "0"])
def test_captured_training_step_two_ranks_keep_identical_parameters(tmp_path, in_graph):
    """DeviceTrainer.capture_step on two ranks (one GPU each, different frames): the gradient all-reduce is captured INSIDE the
    step's hipGraph (in_graph = 1; training.py falls back to two graphs with the all-reduce between them if RCCL refuses to be
    captured, and says so) or issued between two graphs (in_graph = 0, forced).  After three replays both ranks hold identical
    parameters, different from the initial ones, and the two modes agree with each other on rank 0."""
    code = r'''
import os, sys, torch, torch.distributed as dist
root = sys.argv[1]
sys.path.insert(0, os.path.join(root, "second.pytorch_amd")); sys.path.insert(0, root)
from second_amd import distributed as D, synthetic as syn
from second_amd.models import SecondDetector, CAR_FHD
from second_amd.training import DeviceTrainer
import numpy as np
rank, local, world = D.init_from_env("nccl")
torch.cuda.set_device(local)
torch.manual_seed(0)
det = SecondDetector(CAR_FHD).cuda()
tr = DeviceTrainer(det, amp_dtype=torch.bfloat16)
clouds = [syn.syn_kitti_cloud(rank * 2 + s, num_points=6000, num_voxels=5000) for s in range(2)]
boxes = [syn.syn_kitti_boxes(rank * 2 + s, 8) for s in range(2)]
pts, offs = syn.batch_clouds(clouds)
gt = np.concatenate(boxes).astype(np.float32)
goffs = np.cumsum([0] + [len(b) for b in boxes]).astype(np.int32)
ins = [torch.from_numpy(a).cuda() for a in (pts, offs, gt, goffs)]
start = tr.opt.flat.clone()
replay = tr.capture_step(*ins)
for _ in range(3):
    replay()
torch.cuda.synchronize()
flat = tr.opt.flat.clone()
both = [torch.zeros_like(flat) for _ in range(world)]
dist.all_gather(both, flat)
assert torch.equal(both[0], both[1]), "ranks diverged"
assert not torch.equal(flat, start) and torch.isfinite(flat).all()
if rank == 0:
    torch.save(flat.cpu(), sys.argv[2])
    print("CAPTURED_OK in_graph=%d graphs_per_step=%d" % (int(tr.allreduce_in_graph), 1 if tr.allreduce_in_graph else 2))
dist.destroy_process_group()
'''
    path = tmp_path / "captured_ddp.py"
    path.write_text(code)
    out = tmp_path / f"flat_{in_graph}.pt"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", SEC_TRAIN_ALLREDUCE_IN_GRAPH=in_graph)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29631" if in_graph == "1" else "29633", str(path), ROOT, str(out)],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "CAPTURED_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
    if in_graph == "0":
        assert "graphs_per_step=2" in r.stdout
    other = tmp_path.parent / "captured_ddp_ref.pt"
    flat = torch.load(out)
    if other.exists():                                         # second parametrisation: both modes end at the same weights
        diff = (flat - torch.load(other)).abs()               # (atomics in the sparse weight gradient: agreement to rounding, see test_gpu_train_dense)
        assert float((diff > 1e-4).float().mean()) < 2e-3, float(diff.max())
    else:
        torch.save(flat, other)

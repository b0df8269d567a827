"""World-size-2 gloo run (CPU, build container only: needs /root/reference) of the DDP launcher: the reference's UNMODIFIED
``second.pytorch.train.train`` runs three optimisation steps per rank over ``second_amd.launch``'s five seams -- broadcast after
checkpoint restore, ONE flat-bucket gradient all-reduce before ``clip_grad_norm_`` (train.py:323), DistributedSampler on the
training loader, rank-0-only checkpoints / logs / evaluation.  The HIP ops are replaced by the CPU oracle
(tests/oracle_backend.py, test-only); the dataset is synthetic (the container holds no KITTI files) but is built with the
reference's own VoxelGeneratorV2 attributes, ``TargetAssigner.assign`` and ``merge_second_batch`` collate."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

REF = os.environ.get("SECOND_REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "second")), reason="reference checkout not present")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class SynDataset(torch.utils.data.Dataset):
    """Eight synthetic frames in the reference's example format (second/data/preprocess.py:139-400 output keys)."""

    def __init__(self, voxel_generator, target_assigner, out_size_factor, n=8):
        self.vg, self.ta, self.n = voxel_generator, target_assigner, n
        grid = voxel_generator.grid_size
        fm = [*(grid[:2] // out_size_factor), 1][::-1]
        ret = target_assigner.generate_anchors(fm)
        self.anchors = ret["anchors"].reshape(-1, target_assigner.box_ndim)
        self.anchors_dict = target_assigner.generate_anchors_dict(fm)
        self.mt, self.ut = ret["matched_thresholds"], ret["unmatched_thresholds"]
        self.dataset = self                       # train() reaches eval_dataset.dataset.evaluation only on the eval path

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        rng = np.random.default_rng(100 + i)
        r = self.vg.point_cloud_range
        pts = rng.uniform(r[:3] + 0.05, r[3:] - 0.05, (700, 3)).astype(np.float32)
        pts = np.concatenate([pts, rng.uniform(0, 1, (700, 1)).astype(np.float32)], 1)
        vox = self.vg.generate(pts, 2000)
        pick = rng.choice(len(self.anchors), 3, replace=False)
        gt = self.anchors[pick].astype(np.float32).copy()
        gt[:, :2] += rng.normal(0, 0.1, (3, 2)).astype(np.float32)
        t = self.ta.assign(self.anchors, self.anchors_dict, gt, None, gt_classes=np.ones(3, np.int32),
                           gt_names=np.array(["Car"] * 3), matched_thresholds=self.mt, unmatched_thresholds=self.ut,
                           importance=np.ones(3, np.float32))
        return {"voxels": vox["voxels"], "num_points": vox["num_points_per_voxel"], "coordinates": vox["coordinates"],
                "num_voxels": np.array([vox["voxels"].shape[0]], np.int64), "anchors": self.anchors,
                "labels": t["labels"], "reg_targets": t["bbox_targets"], "importance": t["importance"],
                "metrics": {"voxel_gene_time": 0.0, "prep_time": 0.0}, "metadata": {"image_idx": i}}


def _worker(rank, world, port, model_dir, q):
    try:
        _worker_body(rank, world, port, model_dir, q)
    except BaseException as e:   # noqa: BLE001 -- report instead of leaving the parent waiting on the queue
        import traceback
        q.put((rank, "error", traceback.format_exc()))
        raise


def _worker_body(rank, world, port, model_dir, q):
    # SEC_ACCELERATE_TRAIN: the launcher passes the REAL reference network through compat.accelerate_model(train_dtype=bf16).  On this
    # CPU run every training call must be routed to the original forward (the captured step needs the GPU) without disturbing the loop
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      SEC_LAUNCH_NO_ISOLATION="1", SEC_ACCELERATE_TRAIN="bf16")
    for p in (ROOT, os.path.join(ROOT, "second.pytorch_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    torch.set_num_threads(4)
    import oracle_backend
    from second_amd import compat, launch
    compat.install(REF)
    from google.protobuf import text_format
    from second.protos import pipeline_pb2
    from second.pytorch.builder import input_reader_builder
    cfg = pipeline_pb2.TrainEvalPipelineConfig()
    text_format.Merge(open(os.path.join(REF, "second/configs/car.fhd.config")).read(), cfg)
    m = cfg.model.second
    m.voxel_generator.point_cloud_range[:] = [0, -8.0, -3, 17.6, 8.0, 1]          # a 352 x 320 x 40 grid: CPU-sized
    m.target_assigner.class_settings[0].anchor_generator_range.anchor_ranges[:] = [0, -8.0, -1.0, 17.6, 8.0, -1.0]
    m.post_center_limit_range[:] = [0, -8.0, -2.2, 17.6, 8.0, 0.8]
    cfg.train_config.steps = 3                              # the one-cycle schedule needs int(0.4 * steps) >= 1
    cfg.train_config.steps_per_eval = 1000
    cfg.train_input_reader.batch_size = 2
    cfg.train_input_reader.preprocess.num_workers = 0
    cfg.eval_input_reader.preprocess.num_workers = 0
    cfg_path = os.path.join(os.path.dirname(model_dir), f"pipeline_rank{rank}.config")
    with open(cfg_path, "w") as f:
        f.write(text_format.MessageToString(cfg, indent=2))
    built = []

    def build(input_cfg, model_cfg, training, voxel_generator, target_assigner, multi_gpu=False):
        ds = SynDataset(voxel_generator, target_assigner, model_cfg.rpn.layer_strides[0] * 8)
        built.append(training)
        return ds
    input_reader_builder.build = build
    torch.manual_seed(1000 + rank)                          # different initial weights per rank: the broadcast must fix that
    with oracle_backend.installed():
        state = launch.run_train(REF, cfg_path, model_dir, backend="gloo", device=torch.device("cpu"), display_step=1)
    import second.pytorch.train as T
    # after three synchronised steps every rank holds the same parameters
    net = state["net"]
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    import torch.distributed as dist
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    same = all(torch.equal(gathered[0], g) for g in gathered)
    eng = getattr(net, "_second_amd_engine", None)
    assert state.get("train_accelerated") and eng is not None and eng.train_dtype == torch.bfloat16
    assert eng.trainer not in (None, False), eng.stats       # the real network's loss settings are inside the captured step's reach ...
    assert eng.stats["original_calls"] == 3 and eng.stats["train_calls"] == 0, eng.stats      # ... and CPU examples take the original forward
    q.put((rank, same, state["allreduce_calls"], state["allreduce_bytes"], int(net.get_global_step()), built,
           T.torch.utils.data.DataLoader is torch.utils.data.DataLoader))
    dist.destroy_process_group()


def test_reference_train_loop_two_ranks_three_steps(tmp_path):
    world, port = 2, _free_port()
    model_dir = str(tmp_path / "model")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, model_dir, q)) for r in range(world)]
    [p.start() for p in procs]
    res = []
    for _ in range(world):
        r = q.get(timeout=900)
        assert r[1] != "error", r[2]
        res.append(r)
    res.sort()
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    for rank, same, calls, nbytes, gstep, built, _ in res:
        assert same, "ranks diverged"
        assert calls == 3 and nbytes > 1_000_000 and gstep == 3
        assert built == [True, False]
    # rank 0 wrote the checkpoints and the log; nobody else did
    files = sorted(os.listdir(model_dir))
    assert any(f.startswith("voxelnet-") and f.endswith(".tckpt") for f in files), files
    assert "log.txt" in files or any("log" in f for f in files), files
    assert len([f for f in files if f.startswith("voxelnet-")]) == 1


# ------------------------------------------------------------------------------------------ launch.py evaluate
def _eval_worker(rank, world, port, model_dir, q):
    try:
        _eval_body(rank, world, port, model_dir, q)
    except BaseException:   # noqa: BLE001
        import traceback
        q.put((rank, "error", traceback.format_exc()))
        raise


def _eval_body(rank, world, port, model_dir, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      SEC_LAUNCH_NO_ISOLATION="1")
    for p in (ROOT, os.path.join(ROOT, "second.pytorch_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    torch.set_num_threads(4)
    import oracle_backend
    from second_amd import compat, launch
    compat.install(REF)
    from google.protobuf import text_format
    from second.protos import pipeline_pb2
    from second.pytorch.builder import input_reader_builder
    cfg = pipeline_pb2.TrainEvalPipelineConfig()
    text_format.Merge(open(os.path.join(REF, "second/configs/car.fhd.config")).read(), cfg)
    m = cfg.model.second
    m.voxel_generator.point_cloud_range[:] = [0, -8.0, -3, 17.6, 8.0, 1]
    cs = m.target_assigner.class_settings[0]
    cs.anchor_generator_range.anchor_ranges[:] = [0, -8.0, -1.0, 17.6, 8.0, -1.0]
    cs.nms_score_threshold = 0.005                          # random-init heads score ~0.01: keep some boxes to compare
    m.post_center_limit_range[:] = [0, -8.0, -2.2, 17.6, 8.0, 0.8]
    cfg.eval_input_reader.batch_size = 2
    cfg.eval_input_reader.preprocess.num_workers = 0
    cfg_path = os.path.join(os.path.dirname(model_dir), f"eval_w{world}_rank{rank}.config")
    with open(cfg_path, "w") as f:
        f.write(text_format.MessageToString(cfg, indent=2))
    evaluated = []

    def build(input_cfg, model_cfg, training, voxel_generator, target_assigner, multi_gpu=False):
        ds = SynDataset(voxel_generator, target_assigner, model_cfg.rpn.layer_strides[0] * 8, n=7)   # 7: ragged over 2 ranks

        def evaluation(detections, output_dir):               # the reference's dataset.evaluation contract (train.py:535-545)
            evaluated.append(len(detections))
            return {"results": {"official": f"{len(detections)} frames"}, "detail": {}}
        ds.evaluation = evaluation
        # the reference's own wrapper: ``dataset`` is a read-only property there (input_reader_builder.py:44-46)
        return input_reader_builder.DatasetWrapper(ds)
    input_reader_builder.build = build
    torch.manual_seed(7)                                      # no checkpoint in model_dir: every rank builds the same network
    os.makedirs(model_dir, exist_ok=True)
    with oracle_backend.installed():
        proxy = launch.run_evaluate(REF, cfg_path, model_dir, backend="gloo", device=torch.device("cpu"), accelerate=True)
    dets = proxy.last_detections
    eng = proxy.net._second_amd_engine              # net(example) was served by the fused engine (dynamic-shape mode on the CPU)
    assert eng.stats["fused_calls"] > 0
    assert eng.stats["original_calls"] == 0
    summary = [(int(d["metadata"]["image_idx"]), d["box3d_lidar"].float().numpy().round(4).tolist(),
                d["scores"].float().numpy().round(5).tolist()) for d in dets]
    q.put((rank, summary, evaluated))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def _run_eval(world, tmp_path):
    port = _free_port()
    model_dir = str(tmp_path / f"eval_model_w{world}")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_eval_worker, args=(r, world, port, model_dir, q)) for r in range(world)]
    [p.start() for p in procs]
    res = []
    for _ in range(world):
        r = q.get(timeout=900)
        assert r[1] != "error", r[2]
        res.append(r)
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    return sorted(res), model_dir


def test_reference_evaluate_two_ranks_equals_one_rank(tmp_path):
    """``launch.py evaluate``: the reference's unmodified evaluate() (train.py:433-545), every rank running net(example) on frames
    rank, rank + world, ...; the detections gathered in dataset order and handed to the dataset's evaluation ONCE, on rank 0."""
    one, _ = _run_eval(1, tmp_path)
    two, model_dir = _run_eval(2, tmp_path)
    ref = one[0][1]
    assert [s[0] for s in ref] == list(range(7)) and one[0][2] == [7]
    assert sum(len(s[1]) for s in ref) > 0, "no detections at all: the comparison would be empty"
    for rank, summary, evaluated in two:
        assert [s[0] for s in summary] == list(range(7)), "gathered detections are not in dataset order"
        assert evaluated == ([7] if rank == 0 else []), "dataset.evaluation must run on rank 0 only"
        for a, b in zip(summary, ref):
            assert a[0] == b[0]
            np.testing.assert_allclose(np.array(a[1]).reshape(-1, 7), np.array(b[1]).reshape(-1, 7), atol=1e-4)
            np.testing.assert_allclose(a[2], b[2], atol=1e-5)
    # rank 0 wrote result.pkl where evaluate() puts it; the other rank wrote its shard elsewhere
    assert os.path.isfile(os.path.join(model_dir, "eval_results", "step_0", "result.pkl"))
    assert os.path.isdir(os.path.join(model_dir, "eval_results_rank1"))

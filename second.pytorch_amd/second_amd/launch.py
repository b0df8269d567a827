"""One-process-per-GPU (DDP-style) launcher for the reference's UNMODIFIED training script.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        -m second_amd.launch --reference-root /path/to/second.pytorch \\
        train --config_path=second/configs/car.fhd.config --model_dir=/data/model
    python [-m torch.distributed.run ... ] -m second_amd.launch --reference-root /path/to/second.pytorch \\
        evaluate --config_path=second/configs/car.fhd.config --model_dir=/data/model        (run_evaluate: sharded over the ranks)

The reference's only multi-GPU mechanism is single-process nn.DataParallel over padded batches
(second/pytorch/train.py:203-206, second/data/preprocess.py:57-88).  ``train()`` offers no hook between
``build_network`` and its loop and touches ``net.<attr>`` directly (train.py:178-186,285,297,326-329), so the model cannot be
wrapped in DistributedDataParallel.  Instead every rank runs the reference's single-GPU path (``multi_gpu=False``) on its own
GPU and this module patches five seams around it -- no file of the reference is edited:

  1. device isolation   HIP_VISIBLE_DEVICES = LOCAL_RANK before torch touches the GPU: the reference hard-codes ``cuda:0``
                        (train.py:29);
  2. same start         ``torchplus.train.try_restore_latest_checkpoints`` is followed by a broadcast of rank 0's
                        parameters and buffers (so a resumed run and a fresh one both start identical on all ranks);
  3. gradient exchange  ``torch.nn.utils.clip_grad_norm_`` (train.py:323: the first thing after ``backward()``) first averages
                        the gradients of all ranks with ONE RCCL all-reduce of a flat bucket (distributed.GradBucket); with
                        SEC_ACCELERATE_TRAIN=bf16 the network ``build_network`` returns is also passed through
                        ``compat.accelerate_model`` -- forward and backward of every step are two hipGraph replays
                        (second_amd/dropin_train.py) whose gradients are born in that bucket;
  4. data sharding      ``torch.utils.data.DataLoader`` with ``shuffle=True`` (the training loader, train.py:262-270) gets an
                        epoch-advancing DistributedSampler; loaders with workers use the ``spawn`` start method, because the
                        workers call ``spconv.utils.VoxelGeneratorV2.generate`` (second/data/preprocess.py:301-316), which
                        runs on the GPU here, and a HIP context does not survive ``fork()``;
  5. rank-0 side effects  checkpoints (``torchplus.train.save_models``), the model log (``SimpleModelLog``) and the periodic
                        evaluation (``steps_per_eval``) happen on rank 0 only; the other ranks wait in the next all-reduce --
                        the process group is therefore created with a long collective timeout (SEC_DIST_TIMEOUT_S, default
                        4 h: a full nuScenes evaluation outlasts the 10-minute RCCL watchdog default).

BatchNorm statistics stay per rank, as in the reference (DataParallel replicas do not synchronise them either).
"""
import os
import sys
import time


def _isolate_device(env=None):
    """Must run before torch initialises the GPU runtime.  Rank r keeps ONE visible GPU -- entry r of a visible-device list the
    user already set, else physical GPU r -- and from then on addresses it as device 0: LOCAL_RANK becomes "0" (the
    reference hard-codes ``cuda:0``, train.py:29, and ``distributed.init_from_env`` binds RCCL to ``cuda:LOCAL_RANK``), the
    launcher's value stays in LOCAL_RANK_ORIGINAL."""
    env = os.environ if env is None else env
    lr = env.get("LOCAL_RANK")
    if lr is not None and env.get("SEC_LAUNCH_NO_ISOLATION") != "1" and "LOCAL_RANK_ORIGINAL" not in env:
        preset = env.get("HIP_VISIBLE_DEVICES") or env.get("CUDA_VISIBLE_DEVICES")
        if preset:
            ids = [t.strip() for t in preset.split(",") if t.strip()]
            if int(lr) >= len(ids):
                raise SystemExit(f"second_amd.launch: LOCAL_RANK={lr} but only {len(ids)} visible device(s): {preset!r}")
            dev = ids[int(lr)]
        else:
            dev = lr
        env["HIP_VISIBLE_DEVICES"] = dev
        env["CUDA_VISIBLE_DEVICES"] = dev
        env["LOCAL_RANK_ORIGINAL"] = lr
        env["LOCAL_RANK"] = "0"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL needs it on this driver


class _NullLog:
    """SimpleModelLog stand-in for ranks != 0 (second/utils/log_tool.py): same methods, no files."""

    def __init__(self, *a, **k):
        pass

    def __getattr__(self, name):
        return lambda *a, **k: None


def patch_reference(train_module, rank, world, device=None, backend=None):
    """Install seams 2-5 on an imported ``second.pytorch.train`` module.  Returns a dict of the originals (tests restore them)."""
    import torch
    import torch.distributed as dist
    import torch.utils.data as tud
    import torchplus.train as tpt
    from . import distributed as D
    T = train_module
    saved = {"clip": torch.nn.utils.clip_grad_norm_, "DataLoader": tud.DataLoader, "restore": tpt.try_restore_latest_checkpoints,
             "save_models": tpt.save_models, "SimpleModelLog": T.SimpleModelLog, "build_network": T.build_network,
             "convert": T.example_convert_to_torch}
    state = {"net": None, "allreduce_bytes": 0, "allreduce_calls": 0}

    def build_network(*a, **k):
        net = saved["build_network"](*a, **k)
        state["net"] = net
        # SEC_ACCELERATE_TRAIN=bf16|fp16: training-mode net(example) (train.py:306) from the captured device step -- and the
        # periodic evaluation inside train() from the inference graph.  The gradient seam below finds the gradients already packed
        # in the engine's bucket (dropin_train registers it as net._sec_grad_bucket).
        from .dropin import _env_train_dtype
        if _env_train_dtype() is not None:
            from . import compat
            compat.accelerate_model(net, strict=False)
            state["train_accelerated"] = getattr(net, "_second_amd_engine", None) is not None
        return net

    def restore(model_dir, objs, *a, **k):
        r = saved["restore"](model_dir, objs, *a, **k)
        for o in objs:                                   # the network (first call) -- optimizers restore from the same files
            if isinstance(o, torch.nn.Module) and world > 1:
                D.broadcast_parameters(o, 0)
                for b in o.buffers():                    # global_step etc.
                    if not b.is_floating_point():
                        dist.broadcast(b.data, 0)
        return r

    def clip_grad_norm_(parameters, *a, **k):
        net = state["net"]
        if net is not None and world > 1:
            state["allreduce_bytes"] = D.allreduce_gradients(net, average=True)
            state["allreduce_calls"] += 1
        return saved["clip"](parameters, *a, **k)

    class EpochSampler(tud.distributed.DistributedSampler):
        """the reference re-iterates the loader in a ``while True`` (train.py:291-293) without set_epoch: advance it here"""

        def __iter__(self):
            it = super().__iter__()
            self.set_epoch(self.epoch + 1)
            return it

    def DataLoader(dataset, *a, **k):
        if k.get("shuffle") and world > 1 and k.get("sampler") is None:
            k["sampler"] = EpochSampler(dataset, num_replicas=world, rank=rank, shuffle=True)
            k["shuffle"] = False
        if k.get("num_workers", 0) > 0 and k.get("multiprocessing_context") is None:
            k["multiprocessing_context"] = "spawn"       # the workers voxelise on the GPU: no fork after HIP initialisation
        return saved["DataLoader"](dataset, *a, **k)

    def save_models(*a, **k):
        if rank == 0:
            return saved["save_models"](*a, **k)

    def convert(example, dtype=torch.float32, dev=None):
        return saved["convert"](example, dtype, dev if dev is not None else device)

    torch.nn.utils.clip_grad_norm_ = clip_grad_norm_
    tud.DataLoader = DataLoader
    tpt.try_restore_latest_checkpoints = restore
    tpt.save_models = save_models
    T.build_network = build_network
    if device is not None:
        T.example_convert_to_torch = convert
    if rank != 0:
        T.SimpleModelLog = _NullLog
    saved["state"] = state
    return saved


def unpatch_reference(train_module, saved):
    import torch
    import torch.utils.data as tud
    import torchplus.train as tpt
    torch.nn.utils.clip_grad_norm_ = saved["clip"]
    tud.DataLoader = saved["DataLoader"]
    tpt.try_restore_latest_checkpoints = saved["restore"]
    tpt.save_models = saved["save_models"]
    train_module.SimpleModelLog = saved["SimpleModelLog"]
    train_module.build_network = saved["build_network"]
    train_module.example_convert_to_torch = saved["convert"]


def load_config(train_module, config_path, rank):
    """The pipeline config as an object (train() accepts one, train.py:150-158); ranks != 0 never reach the periodic
    checkpoint + evaluation block (train.py:386-423)."""
    from google.protobuf import text_format
    from second.protos import pipeline_pb2
    config = pipeline_pb2.TrainEvalPipelineConfig()
    with open(config_path, "r") as f:
        text_format.Merge(f.read(), config)
    if rank != 0:
        config.train_config.steps_per_eval = 2 ** 31 - 1
    return config


def run_train(reference_root, config_path, model_dir, backend=None, device=None, **train_kwargs):
    """Everything ``main`` does after argument parsing; returns the patch state (all-reduce counters) for tests."""
    from pathlib import Path
    from . import compat, distributed as D
    rank, local_rank, world = D.init_from_env(backend)
    compat.install(reference_root)
    import second.pytorch.train as T
    saved = patch_reference(T, rank, world, device=device, backend=backend)
    try:
        config = load_config(T, config_path, rank)
        if rank != 0:                                    # rank 0 creates (and checks) the model directory first
            deadline = time.time() + float(os.environ.get("SEC_LAUNCH_MODEL_DIR_WAIT_S", 600))
            while not Path(model_dir).exists():
                if time.time() > deadline:
                    raise RuntimeError(f"rank {rank}: model_dir {model_dir!r} was never created by rank 0")
                time.sleep(0.1)
            train_kwargs["resume"] = True
        T.train(config, model_dir, multi_gpu=False, **train_kwargs)
    finally:
        unpatch_reference(T, saved)
    return saved["state"]


class _ShardedEvaluation:
    """Proxy of the reference's dataset object (``eval_dataset.dataset``, train.py:535) for an evaluation sharded over ranks:
    every attribute is the dataset's own, except ``evaluation(detections, output_dir)`` -- it first gathers the ranks' detection
    lists (rank r holds frames r, r + world, ...; ``all_gather_object``), restores the dataset order and runs the reference's
    evaluation on rank 0 only (the other ranks return None, which evaluate() accepts, train.py:537)."""

    def __init__(self, dataset, rank, world):
        self.__dict__.update(_ds=dataset, _rank=rank, _world=world)

    def __getattr__(self, name):
        return getattr(self._ds, name)

    def __len__(self):
        return len(self._ds)

    def evaluation(self, detections, output_dir):
        import torch.distributed as dist
        if self._world > 1:
            parts = [None] * self._world
            dist.all_gather_object(parts, _detections_to_host(detections))
            total = sum(len(p) for p in parts)
            detections = [parts[i % self._world][i // self._world] for i in range(total)]
        self.__dict__["last_detections"] = detections
        if self._rank != 0:
            return None
        return self._ds.evaluation(detections, output_dir)


class _EvalDatasetView:
    """What ``input_reader_builder.build`` returns to evaluate(), with ``.dataset`` answering the sharded proxy.  The
    reference's ``DatasetWrapper`` exposes ``dataset`` as a read-only property (input_reader_builder.py:44-46), so the proxy cannot
    be assigned into it; this view forwards ``len`` / indexing (the DataLoader's contract) and every other attribute to the
    wrapper it stands for."""

    def __init__(self, wrapped, proxy):
        self.__dict__.update(_wrapped=wrapped, dataset=proxy)

    def __len__(self):
        return len(self._wrapped)

    def __getitem__(self, idx):
        return self._wrapped[idx]

    def __getattr__(self, name):
        return getattr(self._wrapped, name)


def _detections_to_host(dets):
    import torch
    return [{k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in d.items()} for d in dets]


def run_evaluate(reference_root, config_path, model_dir, backend=None, device=None, accelerate=True, **eval_kwargs):
    """The reference's UNMODIFIED ``second.pytorch.train.evaluate`` (train.py:433-545) over this package: one process per GPU,
    every rank evaluates frames rank, rank + world, ... of the evaluation set (a sequential sharded sampler on the loader
    evaluate() builds, train.py:485-491), the detections are gathered inside ``eval_dataset.dataset.evaluation`` and the KITTI /
    nuScenes metrics run once on rank 0.  ``accelerate``: ``compat.accelerate_model`` on the network ``build_network`` returns (the fused static-capacity
    pipeline behind ``net(example)``; SEC_ACCELERATE_MODEL=0 keeps the module graph), ``compat.accelerate_nms`` (device-resident
    rotate_nms / nms behind VoxelNet.predict, voxelnet.py:452-455,578-584, for whatever still takes the original forward) and
    ``compat.accelerate_eval`` (rotate_iou_gpu_eval of second/utils/eval.py on sec_rotate_iou_f32).  No collective on the data path.  Returns the proxy dataset (tests read ``last_detections``)."""
    import torch
    import torch.utils.data as tud
    from . import compat, distributed as D
    world_env = int(os.environ.get("WORLD_SIZE", 1))
    rank, local_rank, world = D.init_from_env(backend) if world_env > 1 else (0, 0, 1)
    compat.install(reference_root)
    import second.pytorch.train as T
    if accelerate:
        compat.accelerate_nms()
        compat.accelerate_eval()
    saved = {"DataLoader": tud.DataLoader, "build": T.input_reader_builder.build, "convert": T.example_convert_to_torch,
             "build_network": T.build_network}
    holder = {}
    fuse = accelerate and os.environ.get("SEC_ACCELERATE_MODEL", "1") != "0"

    def build_network(*a, **k):
        # the fused static-capacity pipeline behind net(example) (second_amd.dropin); networks outside it stay as built
        net = saved["build_network"](*a, **k)
        # deferred: evaluate() only collects the per-frame dicts while it loops (train.py:519-524) and first reads them when it hands
        # the list to the dataset's evaluation / pickles it (train.py:537-539) -- the calls return at once and overlap on three lanes
        # (SEC_ACCELERATE_DEFERRED=0: every call synchronous)
        holder["net"] = compat.accelerate_model(net, strict=False, deferred=os.environ.get("SEC_ACCELERATE_DEFERRED", "1") != "0") if fuse else net
        return holder["net"]

    class ShardSampler(tud.Sampler):
        def __init__(self, n):
            self.n = n

        def __iter__(self):
            return iter(range(rank, self.n, world))

        def __len__(self):
            return len(range(rank, self.n, world))

    def DataLoader(dataset, *a, **k):
        if world > 1 and not k.get("shuffle") and k.get("sampler") is None:
            k["sampler"] = ShardSampler(len(dataset))
        if k.get("num_workers", 0) > 0 and k.get("multiprocessing_context") is None:
            k["multiprocessing_context"] = "spawn"
        return saved["DataLoader"](dataset, *a, **k)

    def build(*a, **k):
        ds = saved["build"](*a, **k)
        holder["proxy"] = _ShardedEvaluation(ds.dataset, rank, world)
        return _EvalDatasetView(ds, holder["proxy"])

    def convert(example, dtype=torch.float32, dev=None):
        return saved["convert"](example, dtype, dev if dev is not None else device)

    tud.DataLoader = DataLoader
    T.input_reader_builder.build = build
    T.build_network = build_network
    if device is not None:
        T.example_convert_to_torch = convert
    try:
        if rank != 0 and eval_kwargs.get("result_path") is None:      # result.pkl of the shards must not collide with rank 0's
            eval_kwargs["result_path"] = os.path.join(str(model_dir), f"eval_results_rank{rank}")
        T.evaluate(load_config(T, config_path, 0) if isinstance(config_path, str) else config_path, model_dir, **eval_kwargs)
    finally:
        tud.DataLoader = saved["DataLoader"]
        T.input_reader_builder.build = saved["build"]
        T.example_convert_to_torch = saved["convert"]
        T.build_network = saved["build_network"]
    proxy = holder.get("proxy")
    if proxy is not None:
        proxy.__dict__["net"] = holder.get("net")
    return proxy


def main(argv=None):
    _isolate_device()
    argv = list(sys.argv[1:] if argv is None else argv)
    if len(argv) < 3 or argv[0] != "--reference-root" or argv[2] not in ("train", "evaluate"):
        print(__doc__)
        raise SystemExit("usage: -m second_amd.launch --reference-root DIR {train|evaluate} --config_path=... --model_dir=... [--key=value ...]")
    reference_root = argv[1]
    kwargs = {}
    for tok in argv[3:]:
        if not (tok.startswith("--") and "=" in tok):
            raise SystemExit(f"expected --key=value, got {tok!r}")
        key, val = tok[2:].split("=", 1)
        try:
            import ast
            val = ast.literal_eval(val)
        except (ValueError, SyntaxError):
            pass
        kwargs[key] = val
    config_path, model_dir = kwargs.pop("config_path"), kwargs.pop("model_dir")
    kwargs.pop("multi_gpu", None)
    if argv[2] == "evaluate":
        run_evaluate(reference_root, config_path, model_dir, **kwargs)
        return
    state = run_train(reference_root, config_path, model_dir, **kwargs)
    if int(os.environ.get("RANK", 0)) == 0:
        print(f"[second_amd.launch] {state['allreduce_calls']} gradient all-reduces of {state['allreduce_bytes']} bytes each")


if __name__ == "__main__":
    main()

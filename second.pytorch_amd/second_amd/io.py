"""Data formats on either side of the hot path, so that real KITTI frames and the reference's checkpoints
can be fed to the device-resident path when they are available (this environment holds neither):

  * KITTI velodyne ``.bin`` clouds   -- second/data/kitti_dataset.py:193-205: raw little-endian float32, [N, 4] = (x, y, z,
                                        reflectance); the ``velodyne_reduced`` sibling directory (camera-FOV crop written by
                                        create_data.py) is preferred when it exists;
  * torchplus ``.tckpt`` checkpoints -- torchplus/train/checkpoint.py:52-176: ``torch.save(module.state_dict())`` as
                                        ``<name>-<step>.tckpt`` plus a ``checkpoints.json`` index {latest_ckpt, all_ckpts};
                                        SecondDetector uses the reference's module names, so the state dict loads by key.
"""
import json
import os
from pathlib import Path

import numpy as np
import torch


# ----------------------------------------------------------------------------- KITTI point clouds
def kitti_velodyne_path(velodyne_path, root_path=None, prefer_reduced=True):
    """Resolve ``info["point_cloud"]["velodyne_path"]`` like KittiDataset.get_sensor_data (kitti_dataset.py:193-200)."""
    p = Path(velodyne_path)
    if not p.is_absolute() and root_path is not None:
        p = Path(root_path) / p
    if prefer_reduced:
        reduced = p.parent.parent / (p.parent.stem + "_reduced") / p.name
        if reduced.exists():
            return reduced
    return p


def read_kitti_bin(path, num_features=4):
    """[N, num_features] float32 (kitti_dataset.py:201-204).  Raises on a size that is not a whole number of points."""
    raw = np.fromfile(str(path), dtype=np.float32, count=-1)
    if raw.size % num_features:
        raise ValueError(f"{path}: {raw.size} floats is not a multiple of {num_features} features per point")
    return raw.reshape(-1, num_features)


def write_kitti_bin(path, points):
    np.ascontiguousarray(points, dtype=np.float32).tofile(str(path))


def load_clouds_to_device(paths, device, num_features=4, pin=True):
    """Read several ``.bin`` frames and stage them as ONE device batch for ``SecondDetector.forward_points``:
    (points [sum N, F] float32, point_offsets [B+1] int32).  One pinned staging buffer, one H2D copy."""
    clouds = [read_kitti_bin(p, num_features) for p in paths]
    offs = np.zeros(len(clouds) + 1, np.int32)
    offs[1:] = np.cumsum([c.shape[0] for c in clouds])
    host = torch.empty((int(offs[-1]), num_features), dtype=torch.float32)
    if pin and torch.cuda.is_available():
        host = host.pin_memory()
    for c, lo, hi in zip(clouds, offs[:-1], offs[1:]):
        host[lo:hi] = torch.from_numpy(c)
    return host.to(device, non_blocking=True), torch.from_numpy(offs).to(device, non_blocking=True)


# ----------------------------------------------------------------------------- .tckpt checkpoints
def latest_tckpt(model_dir, model_name="voxelnet"):
    """Path of the newest ``<model_name>-<step>.tckpt`` according to checkpoints.json (checkpoint.py:19-45), else None."""
    info = Path(model_dir) / "checkpoints.json"
    if not info.is_file():
        return None
    with open(info) as f:
        d = json.load(f)
    name = d.get("latest_ckpt", {}).get(model_name)
    if name is None:
        return None
    p = Path(model_dir) / name
    return str(p) if p.is_file() else None


def load_tckpt(path_or_dir, module, model_name="voxelnet", strict=True, map_location="cpu"):
    """Load a reference checkpoint (file, or model directory -> latest) into ``module`` (e.g. SecondDetector).  Returns
    (path, global_step).  ``strict=False`` tolerates the keys SecondDetector does not carry (metrics buffers of VoxelNet)."""
    path = path_or_dir
    if os.path.isdir(path_or_dir):
        path = latest_tckpt(path_or_dir, model_name)
        if path is None:
            raise FileNotFoundError(f"no {model_name} checkpoint listed in {path_or_dir}/checkpoints.json")
    sd = torch.load(path, map_location=map_location)
    own = module.state_dict()
    if not strict:
        sd = {k: v for k, v in sd.items() if k in own and tuple(v.shape) == tuple(own[k].shape)}
    missing = module.load_state_dict(sd, strict=strict)
    if not strict and [k for k in missing.missing_keys if not k.startswith("rpn_acc") and "metrics" not in k]:
        raise KeyError(f"checkpoint {path} lacks parameters of the model: {missing.missing_keys[:5]} ...")
    step = int(sd["global_step"].item()) if "global_step" in sd else int(Path(path).stem.split("-")[-1])
    return str(path), step


def save_tckpt(model_dir, module, global_step, model_name="voxelnet", max_to_keep=8):
    """Write ``<model_name>-<step>.tckpt`` and update checkpoints.json the way torchplus does (checkpoint.py:52-124), so the
    reference's ``try_restore_latest_checkpoints`` picks it up."""
    model_dir = Path(model_dir)
    model_dir.mkdir(parents=True, exist_ok=True)
    info_path = model_dir / "checkpoints.json"
    info = {"latest_ckpt": {}, "all_ckpts": {}}
    if info_path.is_file():
        with open(info_path) as f:
            info = json.load(f)
    name = f"{model_name}-{int(global_step)}.tckpt"
    torch.save(module.state_dict(), model_dir / name)
    info["latest_ckpt"][model_name] = name
    allc = [c for c in info["all_ckpts"].get(model_name, []) if (model_dir / c).is_file() and c != name] + [name]
    while len(allc) > max_to_keep:
        os.remove(model_dir / allc.pop(0))
    info["all_ckpts"][model_name] = allc
    with open(info_path, "w") as f:
        json.dump(info, f, indent=2)
    return str(model_dir / name)

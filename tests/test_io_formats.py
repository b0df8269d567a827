"""Readers for the data formats on either side of the hot path: KITTI velodyne .bin clouds
(second/data/kitti_dataset.py:193-205) and torchplus .tckpt checkpoints (torchplus/train/checkpoint.py:52-176)."""
import json
import os

import numpy as np
import pytest
import torch

REF = os.environ.get("SECOND_REFERENCE", "/root/reference")


def test_kitti_bin_round_trip_and_reduced_directory(tmp_path):
    from second_amd import io, synthetic as syn
    cloud = syn.syn_kitti_cloud(0, num_points=2000, num_voxels=1800)
    (tmp_path / "training" / "velodyne").mkdir(parents=True)
    (tmp_path / "training" / "velodyne_reduced").mkdir(parents=True)
    full = tmp_path / "training" / "velodyne" / "000007.bin"
    io.write_kitti_bin(full, cloud)
    io.write_kitti_bin(tmp_path / "training" / "velodyne_reduced" / "000007.bin", cloud[:500])
    np.testing.assert_array_equal(io.read_kitti_bin(full), cloud)
    # KittiDataset.get_sensor_data prefers the camera-FOV crop when it exists (kitti_dataset.py:196-200)
    p = io.kitti_velodyne_path("training/velodyne/000007.bin", root_path=tmp_path)
    assert p.parent.name == "velodyne_reduced" and io.read_kitti_bin(p).shape == (500, 4)
    assert io.kitti_velodyne_path("training/velodyne/000007.bin", root_path=tmp_path, prefer_reduced=False) == full
    with open(tmp_path / "bad.bin", "wb") as f:
        f.write(b"\0" * 20)
    with pytest.raises(ValueError):
        io.read_kitti_bin(tmp_path / "bad.bin")
    pts, offs = io.load_clouds_to_device([full, p], torch.device("cpu"))
    assert pts.shape == (2500, 4) and offs.tolist() == [0, 2000, 2500]
    np.testing.assert_array_equal(pts[2000:].numpy(), cloud[:500])


def test_tckpt_save_load_and_index(tmp_path):
    from second_amd import io
    from second_amd.models import SecondDetector, CAR_FHD
    torch.manual_seed(0)
    a = SecondDetector(CAR_FHD)
    a.global_step += 1234
    path = io.save_tckpt(tmp_path, a, 1234)
    assert os.path.basename(path) == "voxelnet-1234.tckpt"
    idx = json.load(open(tmp_path / "checkpoints.json"))
    assert idx["latest_ckpt"]["voxelnet"] == "voxelnet-1234.tckpt" and idx["all_ckpts"]["voxelnet"] == ["voxelnet-1234.tckpt"]
    io.save_tckpt(tmp_path, a, 2000)
    assert io.latest_tckpt(tmp_path).endswith("voxelnet-2000.tckpt")
    torch.manual_seed(1)
    b = SecondDetector(CAR_FHD)
    assert not torch.equal(a.rpn.conv_box.weight, b.rpn.conv_box.weight)
    got, step = io.load_tckpt(tmp_path, b)
    assert got.endswith("voxelnet-2000.tckpt") and step == 1234
    for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert ka == kb and torch.equal(va, vb)


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "second")), reason="reference checkout not present")
def test_tckpt_interchanges_with_torchplus(tmp_path):
    """Both directions through the reference's own torchplus.train.save_models / try_restore_latest_checkpoints."""
    from second_amd import compat, io
    compat.install(REF)
    import torchplus.train as tpt
    from second_amd.models import SecondDetector, CAR_FHD
    torch.manual_seed(0)
    a = SecondDetector(CAR_FHD)
    a.name = "voxelnet"
    tpt.save_models(str(tmp_path), [a], 77)                       # the reference writes ...
    torch.manual_seed(1)
    b = SecondDetector(CAR_FHD)
    path, _ = io.load_tckpt(tmp_path, b)                          # ... we read
    assert path.endswith("voxelnet-77.tckpt") and torch.equal(a.rpn.conv_cls.weight, b.rpn.conv_cls.weight)
    io.save_tckpt(tmp_path, b, 99)                                # we write ...
    torch.manual_seed(2)
    c = SecondDetector(CAR_FHD)
    c.name = "voxelnet"
    tpt.try_restore_latest_checkpoints(str(tmp_path), [c])        # ... the reference reads
    assert torch.equal(c.rpn.conv_cls.weight, a.rpn.conv_cls.weight)


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "second")), reason="reference checkout not present")
def test_eval_rotate_iou_is_routed_to_the_device_op(golden):
    """second/utils/eval.py:124,175 -> our sec_rotate_iou_f32 (oracle backend here; -m gpu golden test covers the kernel)."""
    import oracle_backend
    from second_amd import compat
    compat.install(REF)
    ev = compat.accelerate_eval()
    g = golden("rotate_iou")
    with oracle_backend.installed():
        import second_amd.ops as ops
        if not hasattr(ops, "rotate_iou") or not torch.cuda.is_available():
            # CPU container: check the patch is in place and the signature matches the original's
            import inspect
            assert list(inspect.signature(ev.rotate_iou_gpu_eval).parameters)[:3] == ["boxes", "query_boxes", "criterion"]
            assert ev.bev_box_overlap.__globals__["rotate_iou_gpu_eval"] is ev.rotate_iou_gpu_eval
            assert ev.rotate_iou_gpu_eval(np.zeros((0, 5), np.float32), g["qboxes"]).shape == (0, len(g["qboxes"]))
            return

"""spconv.utils (spconv/utils/__init__.py upstream): VoxelGeneratorV2, points_to_voxel and the numpy-facing
NMS helpers the reference imports (second/builder/voxel_builder.py:23-32, second/data/preprocess.py:301-316,
second/core/non_max_suppression/nms_gpu.py:8,16, nms_cpu.py:5-6,14,27).

The arithmetic runs on the MI355X through libsecond_hip.so: numpy arrays are staged to the GPU and back
(that is what the upstream `non_max_suppression` did too).  GPU required; no CPU path.
NOTE: HIP contexts do not survive fork().  `import spconv` therefore makes torch DataLoader workers default to the `spawn`
start method (the reference forks them, train.py:262-277), and a child that WAS forked from a GPU-initialised parent gets a
SecondHipError with instructions from every entry point (second_amd.runtime.check_not_forked) -- never a hang.
"""
import numpy as np
import torch

from second_amd import ops as _ops


def _dev():
    from second_amd.runtime import check_not_forked
    check_not_forked()      # a forked DataLoader worker of a GPU-initialised parent: clear error instead of a hang
    if not torch.cuda.is_available():
        from second_amd.runtime import SecondHipError
        raise SecondHipError("spconv.utils needs a GPU: the MI355X path has no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


class VoxelGeneratorV2:
    def __init__(self, voxel_size, point_cloud_range, max_num_points, max_voxels=20000, full_mean=False,
                 block_filtering=False, block_factor=0, block_size=0, height_threshold=0.0,
                 height_high_threshold=3.0, max_voxels_mode="break"):
        assert full_mean is False, "full_mean is asserted off upstream and unused by every config"
        point_cloud_range = np.array(point_cloud_range, dtype=np.float32)
        voxel_size = np.array(voxel_size, dtype=np.float32)
        grid_size = (point_cloud_range[3:] - point_cloud_range[:3]) / voxel_size
        grid_size = np.round(grid_size).astype(np.int64)
        self._voxel_size = voxel_size
        self._point_cloud_range = point_cloud_range
        self._max_num_points = int(max_num_points)
        self._max_voxels = int(max_voxels)
        self._grid_size = grid_size
        self._full_mean = full_mean
        self._block_filtering = block_filtering
        self._block_factor, self._block_size = block_factor, block_size
        self._height_threshold, self._height_high_threshold = height_threshold, height_high_threshold
        self._mode = max_voxels_mode
        if block_filtering:
            assert block_factor > 0 and block_size > 0, "block_filtering needs block_factor / block_size"

    # -- device-resident path (no host round trip): points [N,F] cuda float32, offsets [B+1] cuda int32
    def generate_device(self, points, point_offsets, max_voxels=None, mean_features=0, sync=True, mean_dtype=None, fill=True):
        if not self._block_filtering:
            return _ops.voxelize(points, point_offsets, self._point_cloud_range.tolist(), self._voxel_size.tolist(),
                                 self._max_num_points, int(max_voxels or self._max_voxels), self._mode,
                                 mean_features=mean_features, sync=sync, mean_dtype=mean_dtype, fill=fill)
        # points_to_voxel_3d_with_filtering: voxelise, then drop flat (ground-only) neighbourhoods
        vox = _ops.voxelize(points, point_offsets, self._point_cloud_range.tolist(), self._voxel_size.tolist(),
                            self._max_num_points, int(max_voxels or self._max_voxels), self._mode, sync=False)
        out = _ops.voxel_block_filter(vox, self._grid_size[:2].tolist(), self._block_factor, self._block_size,
                                      self._height_threshold, self._height_high_threshold, sync=sync)
        if mean_features:
            v = out["voxels"][:, :, :mean_features].sum(1)
            out["mean"] = v / out["num_points_per_voxel"].clamp(min=1).to(v.dtype).unsqueeze(1)
            if mean_dtype is not None:
                out["mean"] = out["mean"].to(mean_dtype)
        return out

    def _run(self, points, max_voxels):
        dev = _dev()
        pts = torch.from_numpy(np.ascontiguousarray(points, dtype=np.float32)).to(dev)
        offs = torch.tensor([0, pts.shape[0]], dtype=torch.int32, device=dev)
        return self.generate_device(pts, offs, max_voxels)

    def generate(self, points, max_voxels=None):
        r = self._run(points, max_voxels)
        n = r["voxel_num"]
        voxels = r["voxels"].cpu().numpy()
        npv = r["num_points_per_voxel"].cpu().numpy()
        mask = (np.arange(self._max_num_points)[None, :] < npv[:, None]).astype(np.float32)[..., None]
        return {"voxels": voxels, "coordinates": r["coordinates"][:, 1:].cpu().numpy(),
                "num_points_per_voxel": npv, "voxel_point_mask": mask, "voxel_num": n}

    def generate_multi_gpu(self, points, max_voxels=None):
        mv = int(max_voxels or self._max_voxels)
        res = self.generate(points, mv)
        n = res["voxel_num"]
        f = points.shape[1]
        out = {"voxels": np.zeros((mv, self._max_num_points, f), np.float32),
               "coordinates": np.zeros((mv, 3), np.int32),
               "num_points_per_voxel": np.zeros((mv,), np.int32),
               "voxel_point_mask": np.zeros((mv, self._max_num_points, 1), np.float32), "voxel_num": n}
        for k in ("voxels", "coordinates", "num_points_per_voxel", "voxel_point_mask"):
            out[k][:n] = res[k]
        return out

    @property
    def voxel_size(self):
        return self._voxel_size

    @property
    def max_num_points_per_voxel(self):
        return self._max_num_points

    @property
    def point_cloud_range(self):
        return self._point_cloud_range

    @property
    def grid_size(self):
        return self._grid_size


VoxelGenerator = VoxelGeneratorV2


def points_to_voxel(points, voxel_size, coors_range, coor_to_voxelidx=None, max_points=35, max_voxels=20000,
                    full_mean=False, block_filtering=False, block_factor=1, block_size=8, height_threshold=0.2,
                    pad_output=False):
    """Functional form (kittiviewer/viewer.py:389-395 is the only caller in the reference; it passes none of the filtering
    arguments).  ``block_filtering`` etc. are forwarded to the generator; ``coor_to_voxelidx`` (upstream's dense scratch
    grid) is accepted and ignored -- the device path hashes."""
    gen = VoxelGeneratorV2(voxel_size, coors_range, max_points, max_voxels, full_mean=full_mean, block_filtering=block_filtering,
                           block_factor=block_factor, block_size=block_size, height_threshold=height_threshold)
    return gen.generate_multi_gpu(points, max_voxels) if pad_output else gen.generate(points, max_voxels)


# ----------------------------------------------------------------------------- NMS helpers (numpy facing)
def non_max_suppression(sorted_dets, keep_out, thresh, device_id=0):
    """spconv's CUDA NMS (src/utils/nms.cu) signature: boxes [N,5] (x1,y1,x2,y2,score) sorted by score,
    writes kept positions into keep_out, returns their count ('+1' convention, IoU > thresh)."""
    dev = _dev()
    if dev.type == "cuda":
        dev = torch.device("cuda", device_id)
    n = sorted_dets.shape[0]
    if n == 0:
        return 0
    d = torch.from_numpy(np.ascontiguousarray(sorted_dets, np.float32)).to(dev).unsqueeze(0)
    cnt = torch.tensor([n], dtype=torch.int32, device=dev)
    keep, num = _chunked_nms(d, cnt, thresh, "axis_aligned", "numba", 1.0)
    k = int(num[0].item())
    keep_out[:k] = keep[0, :k].cpu().numpy()
    return k


def _chunked_nms(dets, counts, thresh, kind, semantics, eps, post_max=0):
    assert dets.shape[1] <= 4096, "sec_nms_sorted_f32 handles up to 4096 boxes per item"
    return _ops.nms_sorted(dets.contiguous(), counts, thresh, kind, semantics, eps, post_max)


def non_max_suppression_cpu(dets, order, thresh, eps=0.0):
    """greedy NMS with the eps convention, IoU >= thresh (nms_cpu.py:11-14); returns original indices."""
    dev = _dev()
    order = np.asarray(order)
    n = dets.shape[0]
    if n == 0:
        return []
    d = torch.from_numpy(np.ascontiguousarray(dets[order], np.float32)).to(dev).unsqueeze(0)
    cnt = torch.tensor([n], dtype=torch.int32, device=dev)
    keep, num = _chunked_nms(d, cnt, thresh, "axis_aligned", "cpu", eps)
    k = int(num[0].item())
    return order[keep[0, :k].cpu().numpy()].tolist()


def _nms_tensor(boxes, scores, pre_max_size, post_max_size, thresh, eps):
    dev = _dev()
    scores, boxes = scores.to(dev).float(), boxes.to(dev).float()
    if pre_max_size is not None and pre_max_size > 0:
        scores, idx = torch.topk(scores, min(pre_max_size, scores.shape[0]))
    else:
        scores, idx = torch.sort(scores, descending=True)
    d = torch.cat([boxes[idx], scores[:, None]], 1).unsqueeze(0).contiguous()
    cnt = torch.tensor([d.shape[1]], dtype=torch.int32, device=dev)
    keep, num = _chunked_nms(d, cnt, thresh, "axis_aligned", "cpu", eps,
                             post_max_size if post_max_size and post_max_size > 0 else 0)
    return idx[keep[0, :int(num[0].item())].long()]


def _corners_to_rbox(corners):
    """[N,4,2] corners in box_np_ops.center_to_corner_box2d order ((-,-),(-,+),(+,+),(+,-) rotated clockwise)
    -> [N,5] (x, y, w, l, r).  Inverse of second/core/box_np_ops.py:405-425."""
    c = np.asarray(corners, np.float32).reshape(-1, 4, 2)
    ctr = c.mean(1)
    ex, ey = c[:, 3] - c[:, 0], c[:, 1] - c[:, 0]
    w, l = np.linalg.norm(ex, axis=1), np.linalg.norm(ey, axis=1)
    r = np.arctan2(-ex[:, 1], ex[:, 0])
    return np.concatenate([ctr, w[:, None], l[:, None], r[:, None]], 1).astype(np.float32)


def rotate_non_max_suppression_cpu(box_corners, order, standup_iou, thresh):
    """spconv's CPU rotated NMS (called by rotate_nms_cc, second/core/non_max_suppression/nms_cpu.py:17-28):
    greedy over `order`, pairs with standup IoU <= 0 skipped, suppress at IoU >= thresh.  Runs on the MI355X
    (the standup matrix is recomputed on the device; the argument is accepted for signature parity)."""
    dev = _dev()
    order = np.asarray(order)
    n = len(order)
    if n == 0:
        return []
    boxes = _corners_to_rbox(box_corners)[order]
    d = torch.from_numpy(np.ascontiguousarray(boxes)).to(dev).unsqueeze(0)
    cnt = torch.tensor([n], dtype=torch.int32, device=dev)
    keep, num = _chunked_nms(d, cnt, thresh, "rotate", "cpu", 0.0)
    return order[keep[0, :int(num[0].item())].cpu().numpy()].tolist()


def rbbox_iou(box_corners, qbox_corners, standup_iou, standup_thresh):
    """[N,K] rotated IoU where standup_iou > standup_thresh else 0 (second/core/box_np_ops.py:10-21)."""
    dev = _dev()
    iou = _ops.rotate_iou(torch.from_numpy(_corners_to_rbox(box_corners)).to(dev),
                          torch.from_numpy(_corners_to_rbox(qbox_corners)).to(dev), -1).cpu().numpy()
    return np.where(np.asarray(standup_iou) > standup_thresh, iou, 0).astype(np.asarray(box_corners).dtype)


def rbbox_intersection(box_corners, qbox_corners, standup_iou, standup_thresh):
    """[N,K] area of the rotated intersection polygon (second/core/box_np_ops.py:23-34 ``rinter_cc`` -> spconv
    ``rbbox_intersection``; the reference's own GPU substitute for ``rinter_cc`` is ``rotate_iou_gpu_eval(..., 2)``, the raw
    intersection: second/utils/eval.py:174-175), zero where ``standup_iou <= standup_thresh``."""
    dev = _dev()
    inter = _ops.rotate_iou(torch.from_numpy(_corners_to_rbox(box_corners)).to(dev),
                            torch.from_numpy(_corners_to_rbox(qbox_corners)).to(dev), 2).cpu().numpy()
    return np.where(np.asarray(standup_iou) > standup_thresh, inter, 0).astype(np.asarray(box_corners).dtype)

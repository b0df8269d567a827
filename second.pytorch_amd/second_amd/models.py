"""Host-side mirror of the reference's VoxelNet forward for the configurations BASELINE.json names.

The GPU box has no /root/reference, so bench.py / smoke() / -m gpu tests cannot import the reference's
model code; this module restates the network *topology* (never the code) with the same module names,
parameter shapes and state_dict keys as second/pytorch/models/{voxelnet,middle,rpn,voxel_encoder}.py, so a
reference ``.tckpt`` loads into it and the unmodified reference model (running over our drop-in ``spconv``)
produces the same numbers (tests/test_dropin_reference.py checks that in the build container).

    SimpleVoxel      voxel_encoder.py:207-225   (fused into the voxeliser's epilogue on the native path)
    SpMiddleFHD      middle.py:111-210          (spconv.SubMConv3d / SparseConv3d stack)
    RPNV2            rpn.py:202-420,468-497     (inference: hand-written MFMA convs, sec_conv2d_nhwc / sec_conv1x1_chain_nhwc;
                                                 training and multi-block RPNs: torch convs)
    predict          voxelnet.py:377-645        (decode -> score filter -> top-k -> rotated NMS -> direction fix)
"""
import math
import os

import numpy as np
import torch
from torch import nn
import torch.nn.functional as F

import spconv
from . import ops


# ------------------------------------------------------------------------------------------ config
CAR_FHD = dict(
    name="car.fhd",
    point_cloud_range=[0, -40, -3, 70.4, 40, 1], voxel_size=[0.05, 0.05, 0.1], max_points_per_voxel=5,
    max_voxels=40000, num_point_features=4,
    middle="SpMiddleFHD", middle_in=4,
    rpn=dict(layer_nums=[5], layer_strides=[1], num_filters=[128], upsample_strides=[1],
             num_upsample_filters=[128], num_input_features=128),
    downsample_factor=8,
    anchor_sizes=[[1.6, 3.9, 1.56]], anchor_ranges=[[0, -40.0, -1.00, 70.4, 40.0, -1.00]], rotations=[0, 1.57],
    matched_thresholds=[0.6], unmatched_thresholds=[0.45], assign_per_class=True,
    num_class=1, num_direction_bins=2, direction_offset=0.0, direction_limit_offset=1.0,
    nms_score_threshold=0.3, nms_pre_max_size=1000, nms_post_max_size=100, nms_iou_threshold=0.01,
    use_rotate_nms=True, post_center_range=[0, -40, -2.2, 70.4, 40, 0.8],
)  # second/configs/car.fhd.config


ALL_PP_LARGEA = dict(
    name="nuscenes/all.pp.largea",
    point_cloud_range=[-50, -50, -10, 50, 50, 10], voxel_size=[0.25, 0.25, 20], max_points_per_voxel=60,
    max_voxels=30000, num_point_features=4,
    vfe="PillarFeatureNet", vfe_filters=[64], middle="PointPillarsScatter", middle_in=64,
    rpn=dict(layer_nums=[3, 5, 5], layer_strides=[2, 2, 2], num_filters=[64, 128, 256],
             upsample_strides=[0.25, 0.5, 1], num_upsample_filters=[128, 128, 128], num_input_features=64),
    downsample_factor=8,
    # anchored classes (car, bus, construction_vehicle, trailer x2 sizes, truck); the other five are `no_anchor`
    anchor_sizes=[[1.95017717, 4.60718145, 1.72270761], [2.94046906, 11.1885991, 3.47030982],
                  [2.73050468, 6.38352896, 3.13312415], [3, 15, 3.8], [2, 3, 3.8], [2.4560939, 6.73778078, 2.73004906]],
    anchor_ranges=[[-50, -50, -0.93897414, 50, 50, -0.93897414], [-50, -50, -0.0715754, 50, 50, -0.0715754],
                   [-50, -50, -0.08168083, 50, 50, -0.08168083], [-50, -50, 0.22228277, 50, 50, 0.22228277],
                   [-50, -50, 0.22228277, 50, 50, 0.22228277], [-50, -50, -0.37937912, 50, 50, -0.37937912]],
    anchor_groups=[[0], [1], [2], [3, 4], [5]],   # sizes belonging to one anchor generator (class)
    # class ids (1-based position in class_settings) of the anchored classes; assign_all with per-anchor thresholds (:269)
    group_class_ids=[1, 2, 3, 4, 5], matched_thresholds=[0.4, 0.5, 0.5, 0.5, 0.5],
    unmatched_thresholds=[0.3, 0.35, 0.35, 0.35, 0.35], assign_per_class=False,
    rotations=[0, 1.57],
    num_class=10, num_direction_bins=2, direction_offset=0.78, direction_limit_offset=0.0,
    nms_score_threshold=0.05, nms_pre_max_size=1000, nms_post_max_size=300, nms_iou_threshold=0.5,
    use_rotate_nms=False, post_center_range=[-59.6, -59.6, -10, 59.6, 59.6, 10],
)  # second/configs/nuscenes/all.pp.largea.config


ALL_FHD_NUSC = dict(
    name="nuscenes/all.fhd",
    point_cloud_range=[-49.6, -49.6, -5, 49.6, 49.6, 3], voxel_size=[0.05, 0.05, 0.2], max_points_per_voxel=1,
    max_voxels=90000, num_point_features=4,
    block_filtering=dict(block_factor=1, block_size=8, height_threshold=0.2),   # ground-block filter
    middle="SpMiddleFHD", middle_in=4,
    rpn=dict(layer_nums=[5], layer_strides=[1], num_filters=[128], upsample_strides=[0.5],   # 0.5 = stride-2 conv
             num_upsample_filters=[128], num_input_features=128),
    downsample_factor=16,
    # car, bicycle, bus, construction_vehicle, motorcycle, pedestrian, traffic_cone, trailer (2 sizes), truck, barrier
    anchor_sizes=[[1.95017719, 4.60718155, 1.72270763], [0.6005891, 1.68452156, 1.27192199],
                  [2.94046903, 11.18859863, 3.47030973], [2.73050475, 6.38352919, 3.13312411],
                  [0.76279479, 2.09973788, 1.44403028], [0.66344887, 0.72564369, 1.75748074],
                  [0.39694518, 0.40359262, 1.06232154], [3.0, 15.0, 3.8], [2.0, 3.0, 3.8],
                  [2.45609379, 6.73778057, 2.73004913], [2.49008846, 0.48578221, 0.98297065]],
    anchor_ranges=[[-49.6, -49.6, z, 49.6, 49.6, z] for z in
                   (-0.93897414, -1.03743017, -0.0715754, -0.08168083, -0.99194854, -0.73911035, -1.27868915,
                    0.22228277, 0.22228277, -0.37937912, -1.27247965)],
    anchor_groups=[[0], [1], [2], [3], [4], [5], [6], [7, 8], [9], [10]],
    rotations=[0, 1.57], group_rotations={5: [0], 6: [0]},   # pedestrian / traffic_cone: one rotation
    group_class_ids=[1, 2, 3, 4, 5, 6, 7, 8, 9, 10], matched_thresholds=[0.4, 0.2, 0.5, 0.4, 0.2, 0.5, 0.5, 0.5, 0.5, 0.3],
    unmatched_thresholds=[0.3, 0.15, 0.35, 0.3, 0.15, 0.35, 0.35, 0.35, 0.35, 0.2], assign_per_class=True,   # :85-295
    num_class=10, num_direction_bins=2, direction_offset=0.78, direction_limit_offset=0.0,
    nms_score_threshold=0.05, nms_pre_max_size=1000, nms_post_max_size=300, nms_iou_threshold=0.5,
    use_rotate_nms=False, post_center_range=[-59.6, -59.6, -10, 59.6, 59.6, 10],
)  # second/configs/nuscenes/all.fhd.config


def anchors_per_location(cfg):
    """target_assigner.num_anchors_per_location (target_assigner.py:249-254): sum over generators of sizes x rotations."""
    groups = cfg.get("anchor_groups") or [[i] for i in range(len(cfg["anchor_sizes"]))]
    return sum(len(g) * len(cfg.get("group_rotations", {}).get(gi, cfg["rotations"])) for gi, g in enumerate(groups))


def anchor_class_ranges(cfg, feature_map_size):
    """Start of every anchor generator's range in the class-major anchor array of generate_anchors (+ the total)."""
    d, h, w = feature_map_size
    groups = cfg.get("anchor_groups") or [[i] for i in range(len(cfg["anchor_sizes"]))]
    begin = [0]
    for gi, g in enumerate(groups):
        begin.append(begin[-1] + len(g) * len(cfg.get("group_rotations", {}).get(gi, cfg["rotations"])) * d * h * w)
    return begin


def grid_size_of(cfg):
    r = np.array(cfg["point_cloud_range"], np.float32)
    v = np.array(cfg["voxel_size"], np.float32)
    return np.round((r[3:] - r[:3]) / v).astype(np.int64)  # (x, y, z)


def generate_anchors(cfg, feature_map_size):
    """[A*D*H*W, 7] anchors ordered (anchor, z, y, x) like target_assigner.generate_anchors
    (second/core/target_assigner.py:169-207 over box_np_ops.create_anchors_3d_range :606-638)."""
    d, h, w = feature_map_size
    out = []
    groups = cfg.get("anchor_groups") or [[i] for i in range(len(cfg["anchor_sizes"]))]
    for gi, grp in enumerate(groups):   # one anchor generator: (size, rotation) major, then z, y, x
        rots = np.array(cfg.get("group_rotations", {}).get(gi, cfg["rotations"]), np.float32)
        rng = np.array(cfg["anchor_ranges"][grp[0]], np.float32)
        zc = np.linspace(rng[2], rng[5], d, dtype=np.float32)
        yc = np.linspace(rng[1], rng[4], h, dtype=np.float32)
        xc = np.linspace(rng[0], rng[3], w, dtype=np.float32)
        a = np.zeros((len(grp), len(rots), d, h, w, 7), np.float32)
        a[..., 0] = xc[None, None, None, None, :]
        a[..., 1] = yc[None, None, None, :, None]
        a[..., 2] = zc[None, None, :, None, None]
        for si, sz in enumerate(grp):
            a[si, ..., 3:6] = np.array(cfg["anchor_sizes"][sz], np.float32)
        a[..., 6] = rots[None, :, None, None, None]
        out.append(a.reshape(-1, 7))
    return np.concatenate(out, 0)


# ------------------------------------------------------------------------------------------ modules
class SimpleVoxel(nn.Module):
    def __init__(self, num_input_features=4):
        super().__init__()
        self.num_input_features = num_input_features

    def forward(self, features, num_voxels, coors=None):
        s = features[:, :, :self.num_input_features].sum(dim=1)
        return (s / num_voxels.type_as(features).view(-1, 1)).contiguous()


class PFNLayer(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.linear = nn.Linear(cin, cout, bias=False)
        self.norm = nn.BatchNorm1d(cout, eps=1e-3, momentum=0.01)


class PillarFeatureNet(nn.Module):
    """One-layer PillarFeatureNet (pointpillars.py:150-237); keys pfn_layers.0.{linear,norm}.  Inference on the
    GPU runs the fused sec_pfn_fwd kernel, training on the GPU sec_pfn_train_fwd / _bwd (ops.PFNTrainFunction: batch statistics,
    backward through the argmax; train_backend = "torch" selects the formulation below); the torch formulation serves CPU
    tests and as the autograd reference of the training kernels."""

    def __init__(self, num_input_features=4, num_filters=(64,), voxel_size=(0.2, 0.2, 4), pc_range=(0, -40, -3, 70.4, 40, 1)):
        super().__init__()
        assert len(num_filters) == 1, "the shipped PointPillars configs use a single PFN layer"
        self.pfn_layers = nn.ModuleList([PFNLayer(num_input_features + 5, num_filters[0])])
        self.vx, self.vy = voxel_size[0], voxel_size[1]
        self.x_offset, self.y_offset = self.vx / 2 + pc_range[0], self.vy / 2 + pc_range[1]

    train_backend = "hip"       # "torch": the reference's materialised [P, T, C] formulation in training (A/B, tests)

    def folded(self):
        """(W^T, scale, shift) of Linear + BatchNorm1d in eval mode; cached until one of the five tensors changes (in-place updates
        and load_state_dict bump their version counters): six element-wise launches per forward otherwise."""
        l = self.pfn_layers[0]
        src = (l.linear.weight, l.norm.weight, l.norm.bias, l.norm.running_mean, l.norm.running_var)
        key = tuple((t.data_ptr(), t._version) for t in src)
        cached = getattr(self, "_folded", None)
        if cached is None or cached[0] != key:
            with torch.no_grad():
                scale = (l.norm.weight * torch.rsqrt(l.norm.running_var + l.norm.eps)).float()
                shift = (l.norm.bias - l.norm.running_mean * scale).float()
                cached = self._folded = (key, (l.linear.weight.detach().t().contiguous().float(), scale.detach().contiguous(),
                                               shift.detach().contiguous()))
        return cached[1]

    def forward_slots(self, points, vox, out_dtype=None, num_dev=None):
        """Inference straight from the voxeliser's point lists (``vox`` = generate_device(points, ..., fill=False)): the
        [P, T, 4] pillar tensor is neither written nor read; bit-identical to :meth:`forward` on the materialised pillars."""
        wt, scale, shift = self.folded()
        return ops.pfn_forward_slots(points, vox, wt, scale, shift, self.vx, self.vy, self.x_offset, self.y_offset,
                                     out_dtype=out_dtype, num_dev=num_dev)

    def forward(self, features, num_voxels, coors, out_dtype=None, num_dev=None):
        l = self.pfn_layers[0]
        if features.is_cuda and not self.training and not torch.is_grad_enabled():
            wt, scale, shift = self.folded()
            return ops.pfn_forward(features.float().contiguous(), num_voxels.int(), coors.int(), wt, scale, shift, self.vx,
                                   self.vy, self.x_offset, self.y_offset, out_dtype=out_dtype, num_dev=num_dev)
        bn = l.norm
        if (features.is_cuda and self.training and bn.training and bn.track_running_stats and bn.affine and l.linear.bias is None
                and self.train_backend == "hip" and ops.pfn_train_supported(features, bn.num_features)
                and features.shape[0] > 0):
            # training on the device: sec_pfn_train_fwd / _bwd (batch statistics, argmax backward), no [P, T, C] tensor
            mom = bn.momentum if bn.momentum is not None else 0.1
            out = ops.PFNTrainFunction.apply(features.contiguous(), num_voxels, coors.int(), l.linear.weight, bn.weight, bn.bias,
                                             bn.running_mean, bn.running_var, bn.eps, mom,
                                             (self.vx, self.vy, self.x_offset, self.y_offset))
            ops.bump_bn_counter(bn)
            return out
        n = num_voxels.to(features.dtype).view(-1, 1, 1)
        xyz = features[:, :, :3]
        mean = xyz.sum(1, keepdim=True) / n
        cx = coors[:, 3].to(features.dtype).view(-1, 1) * self.vx + self.x_offset
        cy = coors[:, 2].to(features.dtype).view(-1, 1) * self.vy + self.y_offset
        dec = torch.cat([features, xyz - mean, (features[:, :, 0] - cx).unsqueeze(-1), (features[:, :, 1] - cy).unsqueeze(-1)], -1)
        t = features.shape[1]
        mask = (torch.arange(t, device=features.device).view(1, -1) < num_voxels.view(-1, 1)).to(features.dtype)
        x = l.linear(dec * mask.unsqueeze(-1))
        x = F.relu(l.norm(x.permute(0, 2, 1)).permute(0, 2, 1))
        return x.max(dim=1)[0]


class PointPillarsScatter(nn.Module):
    """pointpillars.py:420-476: pillars -> [B, C, ny, nx] pseudo image, one launch (sec_pillar_scatter)."""

    def __init__(self, output_shape, num_input_features=64):
        super().__init__()
        self.ny, self.nx, self.nchannels = int(output_shape[2]), int(output_shape[3]), num_input_features

    def forward(self, voxel_features, coords, batch_size, channels_last=False, num_dev=None):
        if torch.is_grad_enabled() and voxel_features.requires_grad:   # training: differentiable scatter
            from spconv.functional import PillarScatterFunction
            return PillarScatterFunction.apply(voxel_features.contiguous(), coords.int().contiguous(), batch_size,
                                               self.ny, self.nx, channels_last)
        return ops.pillar_scatter(voxel_features.contiguous(), coords.int().contiguous(), batch_size, self.ny, self.nx,
                                  channels_last=channels_last, num_dev=num_dev)


def _bn1d(c):
    return nn.BatchNorm1d(c, eps=1e-3, momentum=0.01)


def _bn2d(c):
    return nn.BatchNorm2d(c, eps=1e-3, momentum=0.01)


class SpMiddleFHD(nn.Module):
    """14 sparse conv layers; channel plan 4-16-16-32-32-32-64-...-64; state_dict keys
    ``middle_conv.<i>.weight`` as in the reference."""

    def __init__(self, output_shape, num_input_features=4):
        super().__init__()
        self.sparse_shape = [int(output_shape[1]) + 1, int(output_shape[2]), int(output_shape[3])]
        sub = lambda i, o, key: spconv.SubMConv3d(i, o, 3, bias=False, indice_key=key)
        down = lambda i, o, k, s, p: spconv.SparseConv3d(i, o, k, s, padding=p, bias=False)
        layers = []

        def add(conv, c):
            layers.extend([conv, _bn1d(c), nn.ReLU()])
        add(sub(num_input_features, 16, "subm0"), 16)
        add(sub(16, 16, "subm0"), 16)
        add(down(16, 32, 3, 2, 1), 32)
        add(sub(32, 32, "subm1"), 32)
        add(sub(32, 32, "subm1"), 32)
        add(down(32, 64, 3, 2, 1), 64)
        for _ in range(3):
            add(sub(64, 64, "subm2"), 64)
        add(down(64, 64, 3, 2, [0, 1, 1]), 64)
        for _ in range(3):
            add(sub(64, 64, "subm3"), 64)
        add(down(64, 64, (3, 1, 1), (2, 1, 1), 0), 64)
        self.middle_conv = spconv.SparseSequential(*layers)
        self.fused_chain = True       # static inference: all eight rulebooks from one fused build (SparseSequential.plan_chain)

    def forward(self, voxel_features, coors, batch_size, channels_last=False, num_active_dev=None, site_table=None, bev_sparse=False):
        """``site_table``: the ``site_table`` entry of the ops.voxelize result these (unfiltered) coors come from -- the first
        SubM rulebook then looks its sites up in the voxeliser's hash table instead of hashing them again.
        ``bev_sparse``: return a :class:`SparseBEV` (rows + indices) instead of the dense image when the output grid allows it."""
        x = spconv.SparseConvTensor(voxel_features, coors.int(), self.sparse_shape, batch_size,
                                    num_active_dev=num_active_dev)
        if site_table is not None and x.indices.data_ptr() == coors.data_ptr():
            x.site_table = ((x.indices.data_ptr(), x.indices.shape[0]), site_table)
        if num_active_dev is not None and self.fused_chain:
            x.planned = self.middle_conv.plan_chain(x)       # None: layer-by-layer builds
        x = self.middle_conv(x)
        self.last_overflow_checks = x.overflow_checks
        if channels_last and bev_sparse and x.features.shape[1] == 64 and int(x.spatial_shape[0]) == 2:
            return SparseBEV(x)                      # the RPN's first conv gathers from the rows: no dense image
        if channels_last:
            return x.dense_channels_last_2d()
        d = x.dense()
        n, c, dd, h, w = d.shape
        return d.view(n, c * dd, h, w)


class SparseBEV:
    """The sparse middle's output handed to RPNInference WITHOUT ``.dense()``: rows + (b, z, y, x) indices of a [2, H, W] grid.
    ``dense()`` gives the reference's [B, C * D, H, W] channels_last tensor (middle.py:206-210) for consumers that need it."""

    def __init__(self, sp):
        self.sp = sp
        self.features, self.indices, self.num_dev = sp.features, sp.indices, sp.num_active_dev
        self.batch_size, self.spatial_shape = sp.batch_size, [int(v) for v in sp.spatial_shape]

    def site_map(self):
        m = getattr(self.sp, "site_map_tensor", None)
        if m is not None:                            # written by the fused rulebook chain's tables launch
            return m
        # rows numbered by the sorted build of the last strided layer: the map is that build's bitmap ranks (one launch)
        tbl = getattr(self.sp, "site_bitmap", None)
        if (tbl is not None and tbl[0] == (self.indices.data_ptr(), self.indices.shape[0]) and isinstance(tbl[1][0], str)
                and tbl[1][0] == "sorted"):
            tbl[1][1].record_stream(torch.cuda.current_stream())
            return ops.sparse_site_map_sorted(tbl[1][1], self.indices.shape[0], self.batch_size, self.spatial_shape, num_dev=self.num_dev)
        return ops.sparse_site_map(self.indices, self.batch_size, self.spatial_shape, num_dev=self.num_dev)

    def tile_lists(self, layers, masks=False):
        """(order, counts[, nbr_masks]) of ops.rpn_tile_live for the first ``layers`` RPN convs, computed once per tensor: whoever asks
        first decides where the launch sits (the sparse segment of a staged capture, so that it stays out of the serialised RPN segment)."""
        t = getattr(self, "_tile_lists", None)
        if t is None or t[0] != (layers, bool(masks)):
            t = self._tile_lists = ((layers, bool(masks)), ops.rpn_tile_live(self.site_map(), layers, masks=bool(masks)))
        return t[1]

    def dense(self):
        return self.sp.dense_channels_last_2d()


class PillarBEV:
    """The PillarFeatureNet's rows handed to RPNInference WITHOUT PointPillarsScatter's canvas (pointpillars.py:444-476): rows + (b, z, y, x)
    coordinates of an [ny, nx] grid.  ``dense()`` is the scatter's channels_last image for consumers that need it."""

    def __init__(self, features, coords, batch_size, ny, nx, num_dev=None):
        self.features, self.coords, self.num_dev = features.contiguous(), coords.int().contiguous(), num_dev
        self.batch_size, self.ny, self.nx = int(batch_size), int(ny), int(nx)

    def site_map(self):
        return ops.pillar_site_map(self.coords, self.batch_size, self.ny, self.nx, num_dev=self.num_dev)

    def dense(self):
        return ops.pillar_scatter(self.features, self.coords, self.batch_size, self.ny, self.nx, channels_last=True, num_dev=self.num_dev)


class RPNV2(nn.Module):
    """ZeroPad+Conv3x3+BN+ReLU, layer_num x (Conv3x3+BN+ReLU) per block; deconv (or strided conv) per
    block; three 1x1 heads.  Keys: blocks.<b>.<i>, deblocks.<b>.<i>, conv_cls, conv_box, conv_dir_cls."""

    def __init__(self, num_class=1, layer_nums=(5,), layer_strides=(1,), num_filters=(128,), upsample_strides=(1,),
                 num_upsample_filters=(128,), num_input_features=128, num_anchor_per_loc=2, box_code_size=7,
                 num_direction_bins=2, use_direction_classifier=True):
        super().__init__()
        self._num_anchor_per_loc, self._num_class = num_anchor_per_loc, num_class
        self._box_code_size, self._num_direction_bins = box_code_size, num_direction_bins
        self._use_direction_classifier = use_direction_classifier
        self._upsample_start_idx = len(layer_nums) - len(upsample_strides)
        blocks, deblocks = [], []
        cin = num_input_features
        for i, n in enumerate(layer_nums):
            planes = num_filters[i]
            mods = [nn.ZeroPad2d(1), nn.Conv2d(cin, planes, 3, stride=layer_strides[i], bias=False), _bn2d(planes), nn.ReLU()]
            for _ in range(n):
                mods += [nn.Conv2d(planes, planes, 3, padding=1, bias=False), _bn2d(planes), nn.ReLU()]
            blocks.append(nn.Sequential(*mods))
            if i - self._upsample_start_idx >= 0:
                s = upsample_strides[i - self._upsample_start_idx]
                cup = num_upsample_filters[i - self._upsample_start_idx]
                if s >= 1:
                    s = int(round(s))
                    up = nn.ConvTranspose2d(planes, cup, s, stride=s, bias=False)
                else:
                    s = int(round(1 / s))
                    up = nn.Conv2d(planes, cup, s, stride=s, bias=False)
                deblocks.append(nn.Sequential(up, _bn2d(cup), nn.ReLU()))
            cin = planes
        self.blocks, self.deblocks = nn.ModuleList(blocks), nn.ModuleList(deblocks)
        final = sum(num_upsample_filters) if len(num_upsample_filters) else cin
        self.conv_cls = nn.Conv2d(final, num_anchor_per_loc * num_class, 1)
        self.conv_box = nn.Conv2d(final, num_anchor_per_loc * box_code_size, 1)
        if use_direction_classifier:
            self.conv_dir_cls = nn.Conv2d(final, num_anchor_per_loc * num_direction_bins, 1)

    def forward(self, x):
        ups = []
        for i, blk in enumerate(self.blocks):
            x = blk(x)
            if i - self._upsample_start_idx >= 0:
                ups.append(self.deblocks[i - self._upsample_start_idx](x))
        if ups:
            x = torch.cat(ups, dim=1) if len(ups) > 1 else ups[0]
        a = self._num_anchor_per_loc

        def head(conv, code):
            y = conv(x)
            h, w = y.shape[2:]
            return y.view(-1, a, code, h, w).permute(0, 1, 3, 4, 2).contiguous()
        ret = {"box_preds": head(self.conv_box, self._box_code_size), "cls_preds": head(self.conv_cls, self._num_class)}
        if self._use_direction_classifier:
            ret["dir_cls_preds"] = head(self.conv_dir_cls, self._num_direction_bins)
        return ret


RPN_TRAIN_BACKEND = "hip"


def rpn_forward_mixed(rpn, x, dtype, loss_args=None, loss_terms=False):
    """Training forward of an RPNV2 with 16-bit activations over its fp32 master weights (the DeviceTrainer's amp path).  Every
    Conv2d(128, 128, 3, stride 1) + BatchNorm2d + ReLU triple of the blocks runs on the hand-written kernels -- forward and data
    gradient on k_conv2d_halo_reg, weight gradient on k_conv2d_wgrad3x3, BatchNorm (batch statistics) + ReLU fused
    (ops.Conv3x3Function / ops.BatchNormReluFunction; rpn.py:486-497 trained by train.py:316-322); anything else (strided or
    other-width convs, the deblocks, the 1x1 heads) stays on torch under autocast.  ``models.RPN_TRAIN_BACKEND = "miopen"``: all of it
    on torch (the round-2 path; tests compare the two).  Same arithmetic as RPNV2.forward up to 16-bit rounding of the activations.  (autocast's
    weight-cast cache is off: the function is captured into hipGraphs by DeviceTrainer, and a cached cast made during capture
    would be stale on replay.)
    ``loss_args`` = (labels, reg_targets, anchors, importance, loss_cfg): when the heads run as one stacked convolution and the loss
    kernel has that head shape, the loss is taken from the stacked tensor (ops.HeadsLossFunction) and {"loss", "out6"} comes back
    instead of the three prediction tensors; ``loss_terms``: plus "cls_preds" / "cls_loss" / "loc_loss", the per-anchor fp32 tensors of the
    reference's loss dict (voxelnet.py:299-309), from the same launch."""
    use_hip = RPN_TRAIN_BACKEND == "hip" and x.is_cuda
    x = x.to(dtype).contiguous(memory_format=torch.channels_last)

    def run_block(blk, x):
        mods = list(blk.children())
        i, pad = 0, 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, nn.ZeroPad2d):
                pad = m.padding[0]
                i += 1
                continue
            fused = (use_hip and isinstance(m, nn.Conv2d) and i + 2 < len(mods) and isinstance(mods[i + 1], nn.BatchNorm2d)
                     and isinstance(mods[i + 2], nn.ReLU) and m.bias is None and m.kernel_size == (3, 3) and m.stride == (1, 1)
                     and m.padding[0] + pad == 1 and m.padding[1] + pad == 1 and m.dilation == (1, 1) and m.groups == 1
                     and ops.conv2d_wgrad_supported(m.in_channels, m.out_channels, 3, 1, 1, dtype) and mods[i + 1].training
                     and mods[i + 1].track_running_stats and mods[i + 1].affine)
            if fused:
                bn = mods[i + 1]
                y = ops.Conv3x3Function.apply(x, m.weight)
                mom = bn.momentum if bn.momentum is not None else 0.1
                x = ops.BatchNormReluFunction.apply(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, mom, True)
                ops.bump_bn_counter(bn)
                i, pad = i + 3, 0
                continue
            with torch.autocast("cuda", dtype=dtype, enabled=x.is_cuda, cache_enabled=False):
                if pad:
                    x = F.pad(x, (pad, pad, pad, pad))
                    pad = 0
                x = m(x)
            i += 1
        return x
    def run_deblock(deb, x):
        mods = list(deb.children())
        m = mods[0]
        if (use_hip and len(mods) == 3 and isinstance(m, nn.ConvTranspose2d) and isinstance(mods[1], nn.BatchNorm2d)
                and isinstance(mods[2], nn.ReLU) and m.bias is None and m.kernel_size == (1, 1) and m.stride == (1, 1)
                and m.padding == (0, 0) and m.output_padding == (0, 0) and m.groups == 1 and m.in_channels == 128
                and m.out_channels == 128 and dtype in (torch.bfloat16, torch.float16) and mods[1].training
                and mods[1].track_running_stats and mods[1].affine):
            bn = mods[1]
            y = ops.ConvTranspose1x1Function.apply(x, m.weight)
            mom = bn.momentum if bn.momentum is not None else 0.1
            z = ops.BatchNormReluFunction.apply(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, mom, True)
            ops.bump_bn_counter(bn)
            return z
        with torch.autocast("cuda", dtype=dtype, enabled=x.is_cuda, cache_enabled=False):
            return deb(x)
    ups = []
    for i, blk in enumerate(rpn.blocks):
        x = run_block(blk, x)
        if i - rpn._upsample_start_idx >= 0:
            ups.append(run_deblock(rpn.deblocks[i - rpn._upsample_start_idx], x))
    heads = [(rpn.conv_box, rpn._box_code_size, "box_preds"), (rpn.conv_cls, rpn._num_class, "cls_preds")]
    if rpn._use_direction_classifier:
        heads.append((rpn.conv_dir_cls, rpn._num_direction_bins, "dir_cls_preds"))
    a = rpn._num_anchor_per_loc
    if (use_hip and len(ups) == 1 and ups[0].shape[1] == 128 and dtype in (torch.bfloat16, torch.float16)
            and sum(c.out_channels for c, _, _ in heads) <= 64
            and all(isinstance(c, nn.Conv2d) and c.kernel_size == (1, 1) and c.stride == (1, 1) and c.padding == (0, 0) and c.bias is not None
                    and c.in_channels == 128 for c, _, _ in heads)):
        # the three heads as one 128 -> 64 1x1 convolution (output channels stacked, zero padded) on the hand-written kernels
        tot = sum(c.out_channels for c, _, _ in heads)
        wd = heads[0][0].weight
        pads = getattr(rpn, "_sec_head_pads", None)          # the zero rows of the stacked weight / bias: made once, not filled every step
        if pads is None or pads[0].shape[0] != 64 - tot or pads[0].device != wd.device or pads[0].dtype != wd.dtype:
            pads = (torch.zeros((64 - tot, 128, 1, 1), dtype=wd.dtype, device=wd.device), torch.zeros((64 - tot,), dtype=wd.dtype, device=wd.device))
            rpn._sec_head_pads = pads
        wcat = torch.cat([c.weight for c, _, _ in heads] + [pads[0]], 0)
        bcat = torch.cat([c.bias for c, _, _ in heads] + [pads[1]], 0)
        bins = rpn._num_direction_bins if rpn._use_direction_classifier else 0
        if (loss_args is not None and rpn._box_code_size == 7
                and ops.heads_loss_supported(64, a, rpn._num_class, bins, dtype)):
            # the loss straight from the stacked head tensor, its gradient straight back into it (ops.HeadsLossFunction)
            labels, reg_targets, anchors, importance, loss_cfg = loss_args
            res = ops.HeadsLossFunction.apply(ups[0].contiguous(memory_format=torch.channels_last), wcat, bcat, labels, reg_targets,
                                              anchors, importance, a, rpn._num_class, bins, loss_cfg, bool(loss_terms))
            out = {"loss": res[0], "out6": res[1]}
            if loss_terms:
                out.update(cls_preds=res[2], cls_loss=res[3], loc_loss=res[4])
            return out
        y = ops.Heads1x1Function.apply(ups[0].contiguous(memory_format=torch.channels_last), wcat, bcat)
        ret, c0 = {}, 0
        h, w = y.shape[2:]
        for conv, code, name in heads:
            o = y[:, c0:c0 + conv.out_channels]
            c0 += conv.out_channels
            ret[name] = o.reshape(-1, a, code, h, w).permute(0, 1, 3, 4, 2).contiguous()
        return ret
    with torch.autocast("cuda", dtype=dtype, enabled=x.is_cuda, cache_enabled=False):
        if ups:
            x = torch.cat(ups, dim=1) if len(ups) > 1 else ups[0]

        def head(conv, code):
            y = conv(x)
            h, w = y.shape[2:]
            return y.view(-1, a, code, h, w).permute(0, 1, 3, 4, 2).contiguous()
        ret = {"box_preds": head(rpn.conv_box, rpn._box_code_size), "cls_preds": head(rpn.conv_cls, rpn._num_class)}
        if rpn._use_direction_classifier:
            ret["dir_cls_preds"] = head(rpn.conv_dir_cls, rpn._num_direction_bins)
    return ret


def fold_conv_bn_(seq):
    """In place: fold every (Conv2d|ConvTranspose2d, BatchNorm2d) pair of an nn.Sequential (eval mode)."""
    mods = list(seq.children())
    out = []
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)) and i + 1 < len(mods) and isinstance(mods[i + 1], nn.BatchNorm2d):
            bn = mods[i + 1]
            scale = bn.weight.float() * torch.rsqrt(bn.running_var.float() + bn.eps)
            shift = bn.bias.float() - bn.running_mean.float() * scale
            w = m.weight.float()
            if isinstance(m, nn.ConvTranspose2d):
                w = w * scale.view(1, -1, 1, 1)
            else:
                w = w * scale.view(-1, 1, 1, 1)
            b = shift if m.bias is None else m.bias.float() * scale + shift
            m.weight = nn.Parameter(w.to(m.weight.dtype), requires_grad=False)
            m.bias = nn.Parameter(b.to(m.weight.dtype), requires_grad=False)
            out.append(m)
            i += 2
        else:
            out.append(m)
            i += 1
    return nn.Sequential(*out)


class RPNInference(nn.Module):
    """Inference form of a single-block RPNV2: BatchNorm2d folded into the conv weights (scale) and a float32
    bias, ZeroPad2d merged into the conv padding, the stride-1 1x1 ConvTranspose2d rewritten as a 1x1 conv, the three
    1x1 heads merged into one conv (output channels padded to a multiple of 64).  Every 3x3 conv runs on the
    hand-written MFMA kernel with bias + ReLU fused (sec_conv2d_nhwc), the deblock + heads as one fused kernel
    (sec_conv1x1_chain_nhwc); backend="miopen" keeps torch convs + one fused bias/ReLU pass for A/B runs.
    Same arithmetic as RPNV2.forward (rpn.py:314-331,393-420) up to bf16 rounding of the folded weights."""

    @staticmethod
    def supports(rpn):
        """Every deblock must be expressible as a plain conv: Conv2d(k = s, stride = s) ("upsample" stride < 1), the
        stride-1 1x1 ConvTranspose2d, or ConvTranspose2d(k = s, stride = s), which is a 1x1 conv to s*s*Cout channels
        followed by a depth-to-space rearrangement."""
        for d in rpn.deblocks:
            m = list(d.children())[0]
            if isinstance(m, nn.ConvTranspose2d) and (m.kernel_size != m.stride or m.kernel_size[0] != m.kernel_size[1]
                                                      or m.padding != (0, 0) or m.output_padding != (0, 0)):
                return False
        return len(rpn.blocks) >= 1 and len(rpn.deblocks) >= 1

    def __init__(self, rpn, dtype, backend="hip", gather_first=True):
        """``backend``: "hip" = the hand-written MFMA convs (default), "miopen" = torch convolutions + one fused bias / ReLU pass (the
        phase-1 path: A/B yardstick).  ``gather_first``: the first conv reads the sparse rows through a site map instead of a dense
        image when the shapes allow (False: always the dense image)."""
        super().__init__()
        assert self.supports(rpn)
        self.a, self.codes = rpn._num_anchor_per_loc, (rpn._box_code_size, rpn._num_class, rpn._num_direction_bins)

        def fold(mods):
            out, pad, i = [], 0, 0
            while i < len(mods):
                m = mods[i]
                if isinstance(m, nn.ZeroPad2d):
                    pad = m.padding[0]
                elif isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                    bn = mods[i + 1]
                    assert isinstance(bn, nn.BatchNorm2d) and m.bias is None
                    scale = bn.weight.float() * torch.rsqrt(bn.running_var.float() + bn.eps)
                    bias = bn.bias.float() - bn.running_mean.float() * scale
                    w = m.weight.detach().float()
                    up = 1
                    if isinstance(m, nn.ConvTranspose2d):
                        # [Cin, Cout, s, s] -> 1x1 conv with output channel (dy*s + dx)*Cout + co; depth-to-space afterwards
                        up = m.kernel_size[0]
                        w = (w * scale.view(1, -1, 1, 1)).permute(2, 3, 1, 0).reshape(up * up * w.shape[1], w.shape[0], 1, 1)
                        bias = bias.repeat(up * up)
                        stride, padding = [1, 1], [0, 0]
                    else:
                        stride, padding = list(m.stride), [m.padding[0] + pad, m.padding[1] + pad]
                        w = w * scale.view(-1, 1, 1, 1)
                    w = w.to(dtype).contiguous(memory_format=torch.channels_last)
                    out.append((w, bias.detach().contiguous(), stride, padding, up))
                    pad = 0
                    i += 1
                i += 1
            return out
        # execution plan over a flat layer list: ("c", i) = conv layer i of a block, ("u", i) = deblock conv i whose output
        # is one of the concatenated feature maps
        layers, self.plan = [], []
        up0 = rpn._upsample_start_idx
        for bi, blk in enumerate(rpn.blocks):
            for l in fold(list(blk.children())):
                self.plan.append(("c", len(layers)))
                layers.append(l)
            if bi - up0 >= 0:
                (l,) = fold(list(rpn.deblocks[bi - up0].children()))
                self.plan.append(("u", len(layers)))
                layers.append(l)
        single = len(rpn.blocks) == 1
        self.ws = nn.ParameterList([nn.Parameter(w, requires_grad=False) for w, _, _, _, _ in layers])
        self.bs = nn.ParameterList([nn.Parameter(b, requires_grad=False) for _, b, _, _, _ in layers])
        self.cfgs = [(s, p) for _, _, s, p, _ in layers]
        self.ups = [u for _, _, _, _, u in layers]      # depth-to-space factor after the conv (transposed deblocks)
        heads = [rpn.conv_box, rpn.conv_cls] + ([rpn.conv_dir_cls] if rpn._use_direction_classifier else [])
        self.splits = [h.out_channels for h in heads]
        tot = sum(self.splits)
        padc = (-tot) % 8
        hw = torch.cat([h.weight.detach().float() for h in heads] + [torch.zeros(padc, heads[0].in_channels, 1, 1, device=heads[0].weight.device)], 0)
        hb = torch.cat([h.bias.detach().float() for h in heads] + [torch.zeros(padc, device=heads[0].weight.device)], 0)
        self.head_w = nn.Parameter(hw.to(dtype).contiguous(memory_format=torch.channels_last), requires_grad=False)
        self.head_b = nn.Parameter(hb.contiguous(), requires_grad=False)
        # hand-written MFMA conv (sec_conv2d_nhwc, bias + ReLU fused) when shapes allow
        self.return_views = True   # the fused predict kernels read the head output in place through strides
        self.use_hip = backend == "hip" and dtype in (torch.bfloat16, torch.float16)
        self.packed, self.head_packed = [], None
        if self.use_hip:
            for w in self.ws:
                pk = ops.conv2d_pack_weight(w.detach().contiguous())
                if pk is None:
                    self.use_hip = False
                    break
                self.packed.append(pk)
        if self.use_hip:
            cpad = (-hw.shape[0]) % 64
            hw64 = torch.cat([hw, torch.zeros(cpad, *hw.shape[1:], device=hw.device)], 0).to(dtype).contiguous()
            self.head_cout = hw64.shape[0]
            self.head_packed = ops.conv2d_pack_weight(hw64)
            self.head_b64 = torch.cat([hb, torch.zeros(cpad, device=hb.device)]).contiguous()
            self.use_hip = self.head_packed is not None
        # fp32 networks (the reference's default precision): every 3x3 / s1 / p1 128 -> Cout conv of a single-block RPN on the bf16
        # matrix pipe with split operands (sec_conv2d_nhwc_x3: v = hi + lo, three passes, fp32 accumulation); the 1x1 deblock and
        # heads stay torch fp32 convolutions (9 GFLOP against the blocks' 500)
        self.packed_x3 = None
        if backend == "hip" and dtype == torch.float32 and single and next(rpn.parameters()).is_cuda:
            convs = [i for kind, i in self.plan if kind == "c"]
            if all(tuple(self.ws[i].shape[1:]) == (128, 3, 3) and self.ws[i].shape[0] % 128 == 0 and self.cfgs[i] == ([1, 1], [1, 1])
                   and self.ups[i] == 1 for i in convs):
                pk = {i: ops.conv2d_pack_weight_x3(self.ws[i]) for i in convs}
                if all(v is not None for v in pk.values()):
                    self.packed_x3 = pk
                    self._x3_convs = len(convs) if (convs == list(range(len(convs))) and 2 <= len(convs) <= 8
                                                    and all(self.ws[i].shape[0] == 128 for i in convs)) else 0
                    # the 1x1 deblock + merged heads as ONE split-operand launch (sec_conv1x1_chain_x3) when the shapes allow
                    wl_ = self.ws[-1]
                    self.chain_x3 = None
                    if (self.plan[-1][0] == "u" and tuple(wl_.shape) == (128, 128, 1, 1) and self.cfgs[-1] == ([1, 1], [0, 0])
                            and self.ups[-1] == 1 and hw.shape[0] <= 64):
                        pad = 64 - hw.shape[0]
                        hw64 = torch.cat([hw, torch.zeros(pad, *hw.shape[1:], device=hw.device)], 0).contiguous()
                        hb64 = torch.cat([hb, torch.zeros(pad, device=hb.device)]).contiguous()
                        self.chain_x3 = [ops.conv2d_pack_weight_x3(wl_), ops.conv2d_pack_weight_x3(hw64), hb64]
        # deblock (1x1, stride 1, 128 -> 128) + heads (<= 128 padded channels) run as ONE kernel (sec_conv1x1_chain_nhwc)
        wl = self.ws[-1]
        self.concat_in_place = True      # multi-block RPNs: the deblocks write into the concatenated map (False: torch.cat of their outputs)
        self.pillar_rows_first = True    # a PillarBEV input: the first conv reads the pillar rows through a site map when the shapes allow (False: the scattered canvas)
        self.sparse_input = True   # forward()'s input comes from SparseConvTensor.dense(): all-zero halo tiles skip their MFMA loop (bit-identical)
        # first conv straight from the sparse rows (sec_conv2d_nhwc_gather): 3x3 / s1 / p1 on 2 planes x 64 channels; its weights
        # are packed a second time with the input channels in plane-major order (gather_first=False: always the dense image)
        self.gather_packed = None
        w0 = self.ws[0]
        if (self.use_hip and self.plan[0][0] == "c" and tuple(w0.shape[1:]) == (128, 3, 3) and w0.shape[0] % 128 == 0
                and self.cfgs[0] == ([1, 1], [1, 1]) and self.ups[0] == 1 and gather_first):
            perm = ops.gather_channel_perm(64, 2).to(w0.device)
            self.gather_packed = ops.conv2d_pack_weight(w0.detach()[:, perm].contiguous())
        self.chain_tail = (single and self.use_hip and tuple(wl.shape) == (128, 128, 1, 1) and self.cfgs[-1] == ([1, 1], [0, 0])
                           and self.ups[-1] == 1
                           and self.head_cout in (64, 128))
        # Background tiles (sec_conv2d_nhwc_tiles): with the first conv gathered from the sparse rows the site map is at hand, and a
        # tile of conv j's output that no site can reach (j + 1 steps) holds exactly what the network computes for an EMPTY frame at
        # that position -- any weights.  The empty frame's activations are computed once per map size by the
        # same kernels (:meth:`empty_frame_maps`); the convs then compute only the reachable tiles and copy the others.
        # skip_background = False convolves every tile.
        self.skip_background = True
        # lazy_background: the convs write their live tiles only and read their input's background tiles from the previous layer's
        # empty-frame map (sec_conv2d_nhwc_tiles_lazy) -- no copy of the ~1 450 background tiles (47 MB read + 47 MB written) per layer;
        # the fused 1x1 tail then runs on the last conv's lists whatever the live share (x_live_only)
        self.lazy_background = True
        # lazy_heads (set by SecondDetector around forwards whose consumer is the fused predict): the fused 1x1 tail writes the head
        # tensor's live tiles ONLY and the select / decode kernels read every other tile from the empty frame's head map -- the copy
        # of ~60 % of the [B, H, W, 64] head tensor (22 MB per batch of 8) disappears; forward() then adds preds["lazy_heads"]
        self.lazy_heads = False
        # fused_tail: with lazy heads the last 3x3 conv runs the 1x1 tail in its epilogue (sec_conv2d_nhwc_tiles_tail): one launch less, the
        # conv's output never reaches memory
        self.fused_tail = True
        self.last_live_counts = None       # [convs, B] int32 on the device: live tiles per conv and frame of the last forward (bench / tests)
        self.last_tiles_per_frame = None   # tiles of one frame's map in that forward
        self._empty_maps = {}
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._repack())
        convs = [i for kind, i in self.plan if kind == "c"]
        self.background_convs = len(convs) if (
            self.gather_packed is not None and 2 <= len(convs) <= 8 and convs == list(range(len(convs)))
            and all(tuple(self.ws[i].shape) == (128, 128, 3, 3) and self.cfgs[i] == ([1, 1], [1, 1]) and self.ups[i] == 1 for i in convs)) else 0
        if self.packed_x3 is not None:          # fp32: the same live-tile machinery on the split-operand convs (always the lazy form)
            self.background_convs = getattr(self, "_x3_convs", 0)

    # The packed weight images (MFMA slab order, the gather permutation, the hi | lo pairs of the fp32 form) are derived from the
    # folded parameters at construction: keep them in step with the parameters.  In place, so that captured graphs stay valid.
    def _repack(self):
        with torch.no_grad():
            if self.use_hip:
                for i, w in enumerate(self.ws):
                    self.packed[i].copy_(ops.conv2d_pack_weight(w.detach().contiguous()))
                pad = self.head_cout - self.head_w.shape[0]
                hw64 = torch.cat([self.head_w.detach().float(), torch.zeros(pad, *self.head_w.shape[1:], device=self.head_w.device)], 0)
                self.head_packed.copy_(ops.conv2d_pack_weight(hw64.to(self.head_w.dtype).contiguous()))
                self.head_b64[:self.head_b.numel()].copy_(self.head_b)
                if self.gather_packed is not None:
                    perm = ops.gather_channel_perm(64, 2).to(self.ws[0].device)
                    self.gather_packed.copy_(ops.conv2d_pack_weight(self.ws[0].detach()[:, perm].contiguous()))
            if self.packed_x3 is not None:
                for i, pk in self.packed_x3.items():
                    pk.copy_(ops.conv2d_pack_weight_x3(self.ws[i]))
                if getattr(self, "chain_x3", None) is not None:
                    pad = 64 - self.head_w.shape[0]
                    hw64 = torch.cat([self.head_w.detach().float(), torch.zeros(pad, *self.head_w.shape[1:], device=self.head_w.device)], 0)
                    self.chain_x3[0].copy_(ops.conv2d_pack_weight_x3(self.ws[-1]))
                    self.chain_x3[1].copy_(ops.conv2d_pack_weight_x3(hw64.contiguous()))
                    self.chain_x3[2][:self.head_b.numel()].copy_(self.head_b)
        self._empty_maps.clear()

    def _apply(self, fn, *a, **k):
        super()._apply(fn, *a, **k)
        move = lambda t: fn(t) if isinstance(t, torch.Tensor) else t
        self.packed = [move(t) for t in self.packed]
        for name in ("head_packed", "head_b64", "gather_packed"):
            if getattr(self, name, None) is not None:
                setattr(self, name, move(getattr(self, name)))
        if self.packed_x3 is not None:
            self.packed_x3 = {i: move(t) for i, t in self.packed_x3.items()}
            if getattr(self, "chain_x3", None) is not None:
                self.chain_x3 = [move(t) for t in self.chain_x3]
        self._empty_maps.clear()
        return self

    def empty_frame_maps(self, h, w):
        """Output of every 3x3 conv of the block for a frame WITHOUT sites, channels_last [1, 128, h, w] each, from the kernels the
        forward itself uses (so that a copied tile is bit-identical to a computed one).  Cached per map size; the first call for a
        size must not happen inside a graph capture."""
        # keyed on the folded weights too (version counters, storage, device, dtype): load_state_dict into a prepared detector,
        # .to(device) or a dtype change must not leave maps of the old network behind (they fill every background tile)
        src = [self.ws[i] for i in range(self.background_convs)] + [self.bs[i] for i in range(self.background_convs)] + [self.ws[-1], self.bs[-1], self.head_w, self.head_b]
        key = (int(h), int(w), str(self.ws[0].device), self.ws[0].dtype, tuple((t.data_ptr(), t._version) for t in src))
        if key not in self._empty_maps:
            self._empty_maps.clear()
            assert not torch.cuda.is_current_stream_capturing(), "RPNInference.empty_frame_maps: run one eager forward before capturing"
            dev, dt = self.ws[0].device, self.ws[0].dtype
            with torch.no_grad():
                feat = torch.zeros((1, 64), dtype=dt, device=dev)
                sm = torch.zeros((1, 2, h, w), dtype=torch.int32, device=dev)
                x = ops.conv2d_nhwc_gather(feat, sm, self.gather_packed, self.bs[0], 128, relu=True)
                maps = [x]
                for i in range(1, self.background_convs):
                    x = ops.conv2d_nhwc(x, self.packed[i], self.bs[i], 128, 3, 1, 1, relu=True)
                    maps.append(x)
                if self.chain_tail:      # the fused deblock + heads on the empty frame: background of the tail
                    maps.append(ops.conv1x1_chain(x, self.packed[-1], self.bs[-1], self.head_packed, self.head_b64, self.head_cout))
            self._empty_maps[key] = maps
        return self._empty_maps[key]

    @staticmethod
    def _conv1x1_gemm(x, w, b, relu):
        """1x1 / stride 1 conv of a channels_last fp32 map as ONE GEMM over its [pixels, Cin] matrix (a view), bias in the GEMM's
        epilogue: MIOpen's fp32 1x1 convolution of the RPN tail (deblock 128 -> 128, heads 128 -> 64) ran as a grouped-conv kernel
        plus layout transposes, ~430 us per batch of 8 for 14 GFLOP."""
        n, c, h, wd = x.shape
        x2 = x.permute(0, 2, 3, 1).reshape(n * h * wd, c)                        # a view of the channels_last tensor
        y2 = torch.addmm(b.to(x2.dtype), x2, w.reshape(w.shape[0], c).t().to(x2.dtype))
        if relu:
            y2 = torch.relu_(y2)
        return y2.view(n, h, wd, w.shape[0]).permute(0, 3, 1, 2)                 # channels_last [n, cout, h, w]

    def _conv(self, x, i, sparse_input=False):
        w, b, (s, p) = self.ws[i], self.bs[i], self.cfgs[i]
        if self.use_hip:
            y = ops.conv2d_nhwc(x, self.packed[i], b, w.shape[0], w.shape[2], s[0], p[0], relu=True,
                                sparse_input=sparse_input)
        elif (x.dtype == torch.float32 and x.is_cuda and tuple(w.shape[2:]) == (1, 1) and s == [1, 1] and p == [0, 0]
              and x.is_contiguous(memory_format=torch.channels_last)):
            y = self._conv1x1_gemm(x, w, b, relu=True)
        else:
            y = ops.bias_act_(F.conv2d(x, w, None, s, p), b, relu=True)
        u = self.ups[i]
        if u > 1:   # depth-to-space of the channels-last result: [B,H,W,(dy,dx,co)] -> [B,H*u,W*u,co]
            n, c, h, wd = y.shape
            co = c // (u * u)
            y = y.permute(0, 2, 3, 1).reshape(n, h, wd, u, u * co).permute(0, 1, 3, 2, 4).reshape(n, h * u, wd * u, co)
            y = y.permute(0, 3, 1, 2)        # a channels_last [B,co,H*u,W*u] tensor
        return y

    def empty_frame_maps_x3(self, h, w):
        """(hi, lo) plane pairs of every 3x3 conv's output for a frame WITHOUT sites, from the split-operand kernel itself (a copied or
        lazily read tile is then bit-identical to a computed one).  Cached per map size and weight version; not inside a capture."""
        src = [self.ws[i] for i in self.packed_x3] + [self.bs[i] for i in self.packed_x3]
        key = ("x3", int(h), int(w), str(self.ws[0].device), tuple((t.data_ptr(), t._version) for t in src))
        if key not in self._empty_maps:
            assert not torch.cuda.is_current_stream_capturing(), "RPNInference.empty_frame_maps_x3: run one eager forward before capturing"
            self._empty_maps = {k: v for k, v in self._empty_maps.items() if k[0] != "x3"}
            dev = self.ws[0].device
            with torch.no_grad():
                z = torch.zeros((1, 128, h, w), dtype=torch.bfloat16, device=dev).contiguous(memory_format=torch.channels_last)
                hi, lo, maps = z, z.clone(), []
                for kind, i in self.plan:
                    if kind == "c":
                        hi, lo = ops.conv2d_nhwc_x3(hi, lo, self.packed_x3[i], self.bs[i], self.ws[i].shape[0], relu=True)
                        maps.append((hi, lo))
            self._empty_maps[key] = maps
        return self._empty_maps[key]

    def _forward_x3(self, x):
        """fp32 activations as bf16 (hi, lo) plane pairs through the 3x3 convs; merged back to fp32 for the 1x1 tail.  Given the sparse
        middle's rows (a SparseBEV) the convs run on the tiles a site can reach only, exactly like the 16-bit path: conv j computes
        the tiles within j + 1 steps of a site, reads halo pixels of unwritten tiles from the empty frame's planes, and only the last
        conv materialises its background (the torch 1x1 tail reads the whole map)."""
        lists = None
        if isinstance(x, SparseBEV):
            convs = [i for kind, i in self.plan if kind == "c"]
            if self.background_convs and self.skip_background and convs == list(range(len(convs))):
                sm = x.site_map()
                empty = self.empty_frame_maps_x3(sm.shape[2], sm.shape[3])
                live, self.last_live_counts, nbr = x.tile_lists(self.background_convs, masks=True)
                lists = (live, nbr, empty)
            x = x.dense()
        hi, lo = ops.split_bf16x2(x.float().contiguous(memory_format=torch.channels_last))
        first, ups = self.sparse_input, []
        for kind, i in self.plan:
            if kind == "c" and lists is not None:
                live, nbr, empty = lists
                last = i == self.background_convs - 1
                hi, lo = ops.conv2d_nhwc_x3_tiles(hi, lo, self.packed_x3[i], self.bs[i], self.ws[i].shape[0], live[i], self.last_live_counts[i],
                                                  background=empty[i] if last else None, relu=True,
                                                  nbr_masks=nbr[i] if (i > 0 and nbr is not None) else None,
                                                  background_in=empty[i - 1] if (i > 0 and nbr is not None) else None)
            elif kind == "c":
                hi, lo = ops.conv2d_nhwc_x3(hi, lo, self.packed_x3[i], self.bs[i], self.ws[i].shape[0], relu=True, sparse_input=first)
                first = False
            elif getattr(self, "chain_x3", None) is not None:
                w1, w2, b2 = self.chain_x3            # deblock + heads in one launch, straight from the two planes
                return self._split_heads(ops.conv1x1_chain_x3(hi, lo, w1, self.bs[i], w2, b2, 64, relu1=True))
            else:
                ups.append(self._conv(ops.merge_bf16x2(hi, lo), i))
        f = ups[0] if len(ups) == 1 else torch.cat(ups, dim=1)
        if f.is_cuda and f.is_contiguous(memory_format=torch.channels_last):
            y = self._conv1x1_gemm(f, self.head_w, self.head_b, relu=False)
        else:
            y = ops.bias_act_(F.conv2d(f, self.head_w, None), self.head_b, relu=False)
        return self._split_heads(y)

    def _split_heads(self, y):
        n, _, h, wd = y.shape
        ret, c0 = {}, 0
        for name, sz, code in zip(["box_preds", "cls_preds", "dir_cls_preds"], self.splits, self.codes):
            o = y[:, c0:c0 + sz]
            c0 += sz
            o = o.reshape(n, self.a, code, h, wd).permute(0, 1, 3, 4, 2)   # [B,A,H,W,code] view of the head output
            ret[name] = o if self.return_views else o.contiguous()
        return ret

    def forward(self, x):
        if self.packed_x3 is not None:
            return self._forward_x3(x.dense() if isinstance(x, PillarBEV) else x)
        ups = []
        first = self.sparse_input     # x is the scattered sparse-middle output: mostly empty tiles
        gather = None
        if isinstance(x, SparseBEV):
            if self.gather_packed is not None:
                gather = x
            else:
                x = x.dense()
        if isinstance(x, PillarBEV):
            kind0, i0 = self.plan[0]
            w0, (s0, p0) = self.ws[i0], self.cfgs[i0]
            if (self.use_hip and self.pillar_rows_first and kind0 == "c" and x.features.dtype == w0.dtype and w0.shape[2] == w0.shape[3]
                    and ops.conv2d_rows_supported(x.features.shape[1], w0.shape[0], w0.shape[2], s0[0], p0[0], w0.dtype) and w0.shape[1] == x.features.shape[1]):
                pillars, x = x, None
            else:
                x = x.dense()
        else:
            pillars = None
        live = nbr = fused_heads = None
        catbuf, coff = None, 0
        for kind, i in self.plan:
            if kind == "c" and x is None:
                # the first conv straight from the pillar rows (sec_conv2d_nhwc_rows): no zero fill, no scatter, no 82 MB canvas
                x = ops.conv2d_nhwc_rows(pillars.features, pillars.site_map(), self.packed[i], self.bs[i], self.ws[i].shape[0],
                                         self.ws[i].shape[2], self.cfgs[i][0][0], self.cfgs[i][1][0], relu=True)
                first = False
            elif kind == "c" and gather is not None:
                sm = gather.site_map()
                if self.background_convs and self.skip_background:
                    empty = self.empty_frame_maps(sm.shape[2], sm.shape[3])
                    lazy = self.lazy_background and self.background_convs >= 2
                    if lazy:
                        live, self.last_live_counts, nbr = gather.tile_lists(self.list_layers(), masks=True)
                    else:
                        (live, self.last_live_counts), nbr = gather.tile_lists(self.background_convs), None
                    self.last_tiles_per_frame = int(live.shape[-1])
                    x = ops.conv2d_nhwc_gather(gather.features, sm, self.gather_packed, self.bs[i], self.ws[i].shape[0], relu=True,
                                               tile_order=live[0], live_counts=self.last_live_counts[0], background=None if lazy else empty[0])
                else:
                    x = ops.conv2d_nhwc_gather(gather.features, sm, self.gather_packed, self.bs[i], self.ws[i].shape[0], relu=True)
                gather, first = None, False
            elif kind == "c" and live is not None and nbr is not None and self._tail_in_last_conv(i, x, nbr):
                # the last conv with the 1x1 tail (deblock + heads) in its epilogue: its output tile never leaves LDS (sec_conv2d_nhwc_tiles_tail)
                fused_heads = ops.conv2d_nhwc_tiles_tail(x, self.packed[i], self.bs[i], live[i], self.last_live_counts[i], nbr[i], empty[i - 1],
                                                         self.packed[-1], self.bs[-1], self.head_packed, self.head_b64, self.head_cout)
                x = None
            elif kind == "c" and live is not None and nbr is not None:
                # the last conv keeps its copies unless its consumer is the fused 1x1 tail on the same lists (x_live_only below)
                keep = i == self.background_convs - 1 and not self.chain_tail
                x = ops.conv2d_nhwc_tiles(x, self.packed[i], self.bs[i], 128, live[i], self.last_live_counts[i], empty[i] if keep else None,
                                          relu=True, nbr_masks=nbr[i], background_in=empty[i - 1])
            elif kind == "c" and live is not None:
                x = ops.conv2d_nhwc_tiles(x, self.packed[i], self.bs[i], 128, live[i], self.last_live_counts[i], empty[i], relu=True)
            elif kind == "c":
                x = self._conv(x, i, sparse_input=first)
                first = False
            elif self.chain_tail:
                ups.append(None)                     # single block: the deblock runs fused with the heads below
            elif self._deblocks_write_into_the_concat():
                # multi-block RPN: every deblock deposits its channels straight into the map the heads read (no torch.cat; sec_conv2d_nhwc_into)
                w_, (s_, p_) = self.ws[i], self.cfgs[i]
                if catbuf is None:
                    ho, wo = (x.shape[2] + 2 * p_[0] - w_.shape[2]) // s_[0] + 1, (x.shape[3] + 2 * p_[1] - w_.shape[3]) // s_[1] + 1
                    catbuf = torch.empty((x.shape[0], self._cat_channels, ho, wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
                    coff = 0
                ups.append(ops.conv2d_nhwc_into(x, self.packed[i], self.bs[i], w_.shape[0], w_.shape[2], s_[0], p_[0], True, catbuf, coff))
                coff += w_.shape[0]
            else:
                ups.append(self._conv(x, i))
        lazy_heads = None
        if fused_heads is not None:
            y = fused_heads
            lazy_heads = (nbr[self.background_convs, 1], self._split_heads(empty[self.background_convs]))
        elif self.chain_tail and live is not None:
            last = self.background_convs - 1
            want_lazy = self.lazy_heads and nbr is not None and nbr.shape[0] > self.background_convs
            y = ops.conv1x1_chain(x, self.packed[-1], self.bs[-1], self.head_packed, self.head_b64, self.head_cout,
                                  tile_order=live[last], live_counts=self.last_live_counts[last],
                                  background=None if want_lazy else empty[last + 1], x_live_only=nbr is not None)
            if want_lazy:      # (tile-indexed masks of the layer BEHIND the last conv: bit 4 = the last conv's list holds the tile)
                lazy_heads = (nbr[self.background_convs, 1], self._split_heads(empty[last + 1]))
        elif self.chain_tail:
            y = ops.conv1x1_chain(x, self.packed[-1], self.bs[-1], self.head_packed, self.head_b64, self.head_cout)
        else:
            f = catbuf if catbuf is not None else (ups[0] if len(ups) == 1 else torch.cat(ups, dim=1))    # channels_last in, channels_last out
            if self.use_hip:
                y = ops.conv2d_nhwc(f.contiguous(memory_format=torch.channels_last), self.head_packed, self.head_b64,
                                    self.head_cout, 1, 1, 0, relu=False)
            else:
                y = ops.bias_act_(F.conv2d(f, self.head_w, None), self.head_b, relu=False)
        ret = self._split_heads(y)
        if lazy_heads is not None:
            ret["lazy_heads"] = lazy_heads          # consumed by SecondDetector._predict_fused (ops.predict_select / predict_decode lazy=)
        return ret

    def _tail_in_last_conv(self, i, x, nbr):
        """Conv ``i`` is the last 3x3 conv of a single-block RPN whose consumers are all lazy (lazy heads on the same lists): it then runs
        with the 1x1 tail in its epilogue.  SEC_RPN_FUSED_TAIL=0: the two launches (A/B, bit-identical)."""
        import os
        return (self.fused_tail and i == self.background_convs - 1 and self.chain_tail and self.lazy_heads and self.head_cout == 64
                and nbr.shape[0] > self.background_convs and x is not None and x.shape[1] == 128
                and x.dtype in (torch.bfloat16, torch.float16) and os.environ.get("SEC_RPN_FUSED_TAIL", "1") != "0")

    def _deblocks_write_into_the_concat(self):
        """Several deblocks, each a plain conv (no depth-to-space) of a shape sec_conv2d_nhwc_into serves: they fill one preallocated map."""
        v = getattr(self, "_into_ok", None)
        if v is None:
            us = [i for kind, i in self.plan if kind == "u"]
            v = bool(self.use_hip and self.concat_in_place and len(us) > 1 and all(
                self.ups[i] == 1 and self.cfgs[i][0][0] == self.cfgs[i][0][1] and self.cfgs[i][1][0] == self.cfgs[i][1][1] and self.ws[i].shape[2] == self.ws[i].shape[3]
                and ops.conv2d_into_supported(self.ws[i].shape[1], self.ws[i].shape[0], self.ws[i].shape[2], self.cfgs[i][0][0], self.cfgs[i][1][0], self.ws[i].dtype)
                for i in us))
            self._cat_channels = sum(self.ws[i].shape[0] for i in us)
            self._into_ok = v
        return v

    def list_layers(self):
        """Layers of live-tile lists / masks a forward asks rpn_tile_live for: one per 3x3 conv, plus one when the heads are lazy (its
        tile-indexed masks say which tiles the LAST conv's list -- the fused tail's -- holds)."""
        return self.background_convs + (1 if (self.lazy_heads and self.chain_tail and self.background_convs < 8) else 0)


# ------------------------------------------------------------------------------------------ detector
def limit_period(val, offset, period):
    return val - torch.floor(val / period + offset) * period


class SecondDetector(nn.Module):
    """VoxelNet mirror.  ``forward(example)`` accepts the reference's example dict
    (voxels, num_points, coordinates, anchors; voxelnet.py:339-375) and returns the reference's list of
    prediction dicts; ``forward_points`` is the device-resident fast path (raw clouds in, detections out,
    no host sync until the caller reads the result)."""

    def __init__(self, cfg=CAR_FHD):
        super().__init__()
        self.cfg = cfg
        gs = grid_size_of(cfg)
        self.grid_size = gs
        dense_shape = [1] + gs[::-1].tolist() + [64]
        self.pillars = cfg.get("vfe") == "PillarFeatureNet"
        if self.pillars:
            self.voxel_feature_extractor = PillarFeatureNet(cfg["num_point_features"], cfg["vfe_filters"], cfg["voxel_size"],
                                                           cfg["point_cloud_range"])
            self.middle_feature_extractor = PointPillarsScatter(dense_shape, cfg["middle_in"])
        else:
            self.voxel_feature_extractor = SimpleVoxel(cfg["num_point_features"])
            self.middle_feature_extractor = SpMiddleFHD(dense_shape, cfg["middle_in"])
        # a config adopted from a reference-built VoxelNet (second_amd.dropin) names the anchor count only: its anchors arrive
        # with every example (voxelnet.py:358), so no anchor table is generated here
        a_per_loc = int(cfg.get("num_anchor_per_loc") or anchors_per_location(cfg))
        self.num_anchor_per_loc = a_per_loc
        self.rpn = RPNV2(num_class=cfg["num_class"], num_anchor_per_loc=a_per_loc, num_direction_bins=cfg["num_direction_bins"],
                         **cfg["rpn"])
        self.register_buffer("global_step", torch.LongTensor(1).zero_())
        fm = [1, int(gs[1]) // cfg["downsample_factor"], int(gs[0]) // cfg["downsample_factor"]]
        self.feature_map_size = fm
        if cfg.get("anchor_sizes"):
            self.register_buffer("anchors", torch.from_numpy(generate_anchors(cfg, fm)), persistent=False)
        else:
            self.anchors = None
        self.register_buffer("_arange_p", torch.arange(cfg["nms_post_max_size"], dtype=torch.int32), persistent=False)
        self.register_buffer("post_center_range", torch.tensor(cfg["post_center_range"], dtype=torch.float32),
                             persistent=False)
        self._infer_dtype = None
        self.pfn_slots = True        # PointPillars inference: the PillarFeatureNet walks the voxeliser's point lists (False: pillar tensor)
        self.fused_predict = True
        self.rulebook_numbering = os.environ.get("SEC_RULEBOOK_NUMBERING", "sorted")
        bf = cfg.get("block_filtering")
        self.voxel_generator = spconv.utils.VoxelGeneratorV2(cfg["voxel_size"], cfg["point_cloud_range"],
                                                            cfg["max_points_per_voxel"], cfg["max_voxels"],
                                                            **(dict(bf, block_filtering=True) if bf else {}))

    # -- inference preparation: bf16 channels-last RPN with folded BN; sparse stack in bf16 (BN folded at run time)
    def prepare_inference(self, dtype=torch.bfloat16, rpn_backend="hip", gather_first=True, exact=False):
        """``dtype=torch.float32`` selects the fp32-STORAGE pipeline, whose products by default run as three bf16 MFMA passes on split
        operands (16 significant bits per operand, fp32 accumulation; reported as ``ops.FP32_SPLIT_LABEL`` = "bf16x3").  ``exact=True``:
        true fp32 arithmetic, the reference's default precision (train.py:232-235) -- sparse convs on v_mfma_f32_32x32x2_f32 / VALU
        (``ops.fp32_mode("exact")`` around every forward), the RPN on torch's fp32 convolutions: several times slower, for parity work."""
        # NOTE: MIOpen's fused conv+bias+ReLU plan (torch.miopen_convolution_relu) was measured at ~160 ms per
        # 3x3 conv for bf16 NHWC on gfx950 (naive fallback kernel) vs 0.14 ms unfused -- not an option; the
        # fused dense path is the hand-written MFMA conv.
        self.eval()
        self.fp32_exact = bool(exact) and dtype == torch.float32
        if self.fp32_exact:
            rpn_backend = "miopen"
        if isinstance(self.rpn, RPNV2) and RPNInference.supports(self.rpn) and next(self.parameters()).is_cuda:
            self.rpn = RPNInference(self.rpn, dtype, backend=rpn_backend, gather_first=gather_first)
        else:
            self.rpn.blocks = nn.ModuleList([fold_conv_bn_(b) for b in self.rpn.blocks])
            self.rpn.deblocks = nn.ModuleList([fold_conv_bn_(b) for b in self.rpn.deblocks])
            self.rpn.to(dtype).to(memory_format=torch.channels_last)
        for m in self.middle_feature_extractor.modules():
            if isinstance(m, spconv.SparseConvolution):
                m.weight.data = m.weight.data.to(dtype)
        self._infer_dtype = dtype
        return self

    # -- stages ------------------------------------------------------------------------------------
    def network_forward(self, voxel_features, coors, batch_size, num_active_dev=None, site_table=None):
        with ops.fp32_mode("exact" if getattr(self, "fp32_exact", False) else None):
            return self._network_forward(voxel_features, coors, batch_size, num_active_dev, site_table)

    def arithmetic(self):
        """What the prepared pipeline computes with, for records: "bf16" / "fp16" (16-bit features, fp32 accumulation), "fp32" (exact
        mode: IEEE fp32 products) or "bf16x3" (fp32 storage, split-operand products)."""
        dt = self._infer_dtype
        if dt in (torch.bfloat16, torch.float16):
            return "bf16" if dt == torch.bfloat16 else "fp16"
        if getattr(self, "fp32_exact", False) or (next(self.parameters()).is_cuda and ops.get_fp32_mode() == "exact"):
            return "fp32"
        return ops.FP32_SPLIT_LABEL if next(self.parameters()).is_cuda else "fp32"

    def _network_forward(self, voxel_features, coors, batch_size, num_active_dev=None, site_table=None):
        dt = self._infer_dtype
        if self.pillars:
            mfe = self.middle_feature_extractor
            if (dt in (torch.bfloat16, torch.float16) and isinstance(self.rpn, RPNInference) and self.rpn.use_hip and self.rpn.packed_x3 is None
                    and isinstance(mfe, PointPillarsScatter) and not torch.is_grad_enabled()):
                # no canvas: the RPN's first conv gathers the pillar rows (PillarBEV.dense() is the scatter for the shapes it does not take)
                return self.rpn(PillarBEV(voxel_features.to(dt), coors, batch_size, mfe.ny, mfe.nx, num_dev=num_active_dev))
            spatial = mfe(voxel_features if dt is None else voxel_features.to(dt), coors,
                          batch_size, channels_last=dt is not None, num_dev=num_active_dev)
            return self.rpn(spatial)
        if dt is not None:
            spatial = self.middle_feature_extractor(voxel_features.to(dt), coors, batch_size, channels_last=True,
                                                    num_active_dev=num_active_dev, site_table=site_table,
                                                    bev_sparse=self._rpn_takes_rows())
        else:
            spatial = self.middle_feature_extractor(voxel_features, coors, batch_size, num_active_dev=num_active_dev,
                                                    site_table=site_table)
        return self.rpn(spatial)

    def _rpn_takes_rows(self):
        """The RPN consumes the sparse middle's rows + site map (SparseBEV) instead of the dense image: the 16-bit gathered first conv, or
        the fp32 split-operand convs on live tiles."""
        return getattr(self.rpn, "gather_packed", None) is not None or (
            getattr(self.rpn, "packed_x3", None) is not None and getattr(self.rpn, "background_convs", 0) > 0)

    def forward(self, example):
        voxels, num_points, coors = example["voxels"], example["num_points"], example["coordinates"]
        batch_size = example["anchors"].shape[0]
        with torch.no_grad():
            feats = self.voxel_feature_extractor(voxels, num_points, coors)
        preds = self.network_forward(feats, coors, batch_size)
        with torch.no_grad():
            return self.predict(preds, example["anchors"].view(batch_size, -1, 7))

    def forward_points(self, points, point_offsets, static=False):
        """points [N,4] cuda float32 (clouds concatenated), point_offsets [B+1] cuda int32.

        ``static=True``: static-capacity, sync-free pipeline (every buffer sized for N rows, live counts stay
        on the device) -- hipGraph-capturable; call :meth:`check_overflow` whenever the host next syncs.

        The strided rulebooks of this path use ``self.rulebook_numbering`` ("sorted" = spconv's GPU output numbering: no
        hash table; the row order is internal here -- the dense feature map and the detections are identical to the
        first-touch form; SEC_RULEBOOK_NUMBERING overrides)."""
        prev = ops.set_rulebook_numbering(self.rulebook_numbering)
        try:
            return self._forward_points(points, point_offsets, static)
        finally:
            ops.set_rulebook_numbering(prev)

    def lazy_heads(self):
        """Context: forwards inside may leave the head tensor's background tiles unwritten (RPNInference.lazy_heads) because their
        preds go straight to :meth:`predict_device`'s fused kernels, which read those tiles from the empty frame's map.  A no-op when
        the prepared RPN / the predict path cannot do it (or SEC_RPN_LAZY_HEADS=0)."""
        det = self

        class _Ctx:
            def __enter__(self_):
                self_.prev = getattr(det.rpn, "lazy_heads", None)
                if (self_.prev is not None and det.fused_predict and det.cfg["nms_pre_max_size"] <= 1024
                        and os.environ.get("SEC_RPN_LAZY_HEADS", "1") != "0"):
                    det.rpn.lazy_heads = True
                return self_

            def __exit__(self_, *exc):
                if self_.prev is not None:
                    det.rpn.lazy_heads = self_.prev
                return False
        return _Ctx()

    def _forward_points(self, points, point_offsets, static=False):
        with self.lazy_heads():
            return self._forward_points_impl(points, point_offsets, static)

    def _forward_points_impl(self, points, point_offsets, static=False):
        batch_size = point_offsets.numel() - 1
        nf = self.cfg["num_point_features"]
        if self.pillars:
            # inference on 4-feature points: the PillarFeatureNet walks the voxeliser's point lists (no [P, 60, 4] tensor: 98 MB
            # written and re-read per step at nuScenes size); pfn_slots = False materialises the pillars as the reference does
            # (block filtering returns compacted pillars without the voxeliser's point lists: tensor path)
            slots = (not self.training and not torch.is_grad_enabled() and nf == 4 and points.shape[1] == 4 and points.is_cuda
                     and not getattr(self.voxel_generator, "_block_filtering", False)
                     and self.pfn_slots)
            vox = self.voxel_generator.generate_device(points, point_offsets, sync=not static, fill=not slots)
            nd = vox["voxel_offsets"][batch_size:] if static else None    # device count of live pillars
            if slots:
                feats = self.voxel_feature_extractor.forward_slots(points, vox, out_dtype=self._infer_dtype, num_dev=nd)
            else:
                feats = self.voxel_feature_extractor(vox["voxels"], vox["num_points_per_voxel"], vox["coordinates"],
                                                     out_dtype=self._infer_dtype, num_dev=nd)
            preds = self.network_forward(feats, vox["coordinates"], batch_size, num_active_dev=nd)
            return self.predict_device(preds, batch_size)
        if not static:
            vox = self.voxel_generator.generate_device(points, point_offsets, mean_features=nf, mean_dtype=self._infer_dtype)
            preds = self.network_forward(vox["mean"], vox["coordinates"], batch_size, site_table=vox.get("site_table"))
        else:
            # (the voxel means are stored in the sparse stack's dtype by the voxeliser itself: no cast launch inside the captured step)
            vox = self.voxel_generator.generate_device(points, point_offsets, mean_features=nf, sync=False, mean_dtype=self._infer_dtype)
            preds = self.network_forward(vox["mean"], vox["coordinates"], batch_size,
                                         num_active_dev=vox["voxel_offsets"][batch_size:], site_table=vox.get("site_table"))
        return self.predict_device(preds, batch_size)

    def calibrate(self, points, point_offsets, margin=1.25):
        """Size the static-capacity buffers of the strided layers from one eager forward of representative
        clouds (live outputs x margin, rounded up to 256 rows), like a static-shape inference engine profile.
        Data that later exceeds a capacity is reported by :meth:`check_overflow`."""
        with torch.no_grad():
            self.forward_points(points, point_offsets)
        caps = []
        for m in self.middle_feature_extractor.modules():
            if isinstance(m, spconv.SparseConvolution) and not m.subm and m.last_num_out is not None:
                m.static_out_rows = int(-(-int(m.last_num_out * margin) // 256) * 256)
                caps.append(m.static_out_rows)
        return caps

    def check_overflow(self):
        """Raise if a strided layer of the last static forward (every branch of a branched graph) produced more
        outputs than its capacity."""
        checks = list(getattr(self.middle_feature_extractor, "last_overflow_checks", []))
        for lst in getattr(self, "_branch_overflow", []):
            checks += list(lst)
        for num, cap in checks:
            raw = int(num[1].item())
            if raw > cap:
                raise RuntimeError(f"static-capacity overflow: a strided sparse conv produced {raw} outputs, capacity {cap}")

    def make_graphed(self, points, point_offsets, warmup=3, branches=1, parts=None):
        """Capture the static forward into a hipGraph.  Returns (replay_fn, outputs); new clouds are fed by
        copying into ``points`` / ``point_offsets`` (any point count <= capacity) before calling replay_fn.

        ``branches`` > 1: the frames are split into that many contiguous groups, each captured as an independent
        branch of the SAME graph on its own stream (frames are independent).  Most kernels of the path are
        latency bound, so two half-batch chains overlap better than one full-batch chain (1.43 -> 1.33 ms for 8
        frames; 4 branches no better, 8 worse).  Returns (replay_fn, [outputs per branch], [(points_i, offsets_i)]):
        the per-branch input buffers are the ones to refill."""
        if branches <= 1:
            return self._capture([(points, point_offsets)], warmup)[:2]
        if parts is None:      # (``parts``: input buffers of an earlier capture, re-used so that several graphs read the same ones)
            offs = point_offsets.cpu().tolist()
            nfr = len(offs) - 1
            branches = min(branches, nfr)
            bounds = [round(i * nfr / branches) for i in range(branches + 1)]
            parts = []
            for lo, hi in zip(bounds[:-1], bounds[1:]):
                pts = points[offs[lo]:offs[hi]].clone()
                po = torch.tensor([o - offs[lo] for o in offs[lo:hi + 1]], dtype=torch.int32, device=points.device)
                parts.append((pts, po))
        # capacities: the largest any branch needs
        caps = None
        for pts, po in parts:
            c = self.calibrate(pts, po)
            caps = c if caps is None else [max(a, b) for a, b in zip(caps, c)]
        if caps:
            mods = [m for m in self.middle_feature_extractor.modules()
                    if isinstance(m, spconv.SparseConvolution) and not m.subm and m.last_num_out is not None]
            for m, c in zip(mods, caps):
                m.static_out_rows = c
        replay, outs, _ = self._capture(parts, warmup)
        return replay, outs, parts

    def _capture(self, parts, warmup):
        mfe = self.middle_feature_extractor
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s), torch.no_grad():
            for _ in range(warmup):
                for pts, po in parts:
                    self.forward_points(pts, po, static=True)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        outs, self._branch_overflow = [], []
        # thread_local: a RCCL watchdog / other host thread touching the runtime must not invalidate the capture
        with ops.rt.capture_guard(), torch.no_grad(), torch.cuda.graph(graph, capture_error_mode="thread_local"):
            cur = torch.cuda.current_stream()
            side = [torch.cuda.Stream() for _ in parts[1:]]
            for st in side:
                st.wait_stream(cur)
            for st, (pts, po) in zip([cur] + side, parts):
                with torch.cuda.stream(st):
                    outs.append(self.forward_points(pts, po, static=True))
                    self._branch_overflow.append(list(getattr(self.middle_feature_extractor, "last_overflow_checks", [])))
            for st in side:
                cur.wait_stream(st)
        return graph.replay, (outs[0] if len(parts) == 1 else outs), graph

    def _capture_stages(self, points, point_offsets, warmup=3):
        """The static forward as THREE hipGraphs sharing one memory pool -- (A) voxelise + sparse middle, (B) the dense RPN, (C)
        decode / top-k / NMS -- so that a serving loop with several steps in flight can pass a token between the lanes' (B)
        segments (InFlightRunner(serialize_rpn=True)): the MFMA-bound RPN segments then run one after another while the
        latency-bound (A) / (C) segments of the other lanes run beside them.  Returns ((replay_a, replay_b, replay_c), outputs)."""
        assert not self.pillars and self._infer_dtype is not None, "staged capture: the sparse-middle inference path"
        batch_size = point_offsets.numel() - 1
        nf = self.cfg["num_point_features"]
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s), torch.no_grad():
            for _ in range(warmup):
                self.forward_points(points, point_offsets, static=True)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        pool = torch.cuda.graph_pool_handle()
        ga, gb, gc = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        prev = ops.set_rulebook_numbering(self.rulebook_numbering)
        try:
            with torch.no_grad():
                with ops.rt.capture_guard(), torch.cuda.graph(ga, pool=pool, capture_error_mode="thread_local"):
                    vox = self.voxel_generator.generate_device(points, point_offsets, mean_features=nf, sync=False,
                                                               mean_dtype=self._infer_dtype)
                    spatial = self.middle_feature_extractor(vox["mean"].to(self._infer_dtype), vox["coordinates"], batch_size,
                                                            channels_last=True, num_active_dev=vox["voxel_offsets"][batch_size:],
                                                            site_table=vox.get("site_table"),
                                                            bev_sparse=self._rpn_takes_rows())
                    self._branch_overflow = [list(getattr(self.middle_feature_extractor, "last_overflow_checks", []))]
                    if isinstance(spatial, SparseBEV) and getattr(self.rpn, "background_convs", 0) and self.rpn.skip_background:
                        lazy_ = self.rpn.lazy_background and self.rpn.background_convs >= 2
                        with self.lazy_heads():
                            spatial.tile_lists(self.rpn.list_layers() if lazy_ else self.rpn.background_convs,   # the live-tile lists belong to the latency-bound segment
                                               masks=lazy_)
                with ops.rt.capture_guard(), torch.cuda.graph(gb, pool=pool, capture_error_mode="thread_local"):
                    with self.lazy_heads():
                        preds = self.rpn(spatial)
                with ops.rt.capture_guard(), torch.cuda.graph(gc, pool=pool, capture_error_mode="thread_local"):
                    out = self.predict_device(preds, batch_size)
        finally:
            ops.set_rulebook_numbering(prev)
        # buffers the later graphs read: owned by whoever keeps the replays (InFlightRunner stores them per lane); the detector
        # only remembers the LAST capture, so rebuilding runners does not accumulate graph-pool buffers here
        self._last_stage_buffers = (vox, spatial, preds)
        return (ga.replay, gb.replay, gc.replay), out

    # -- post-processing -----------------------------------------------------------------------------
    def _select(self, preds, batch_size, anchors):
        """score filter + top-k + decode of the selected boxes, all on device, fixed shapes."""
        cfg = self.cfg
        nc = cfg["num_class"]
        box = preds["box_preds"].reshape(batch_size, -1, 7)
        if nc == 1:
            scores = torch.sigmoid(preds["cls_preds"].reshape(batch_size, -1).float())
            labels = None
        else:   # class-agnostic selection: best class per anchor (voxelnet.py:545-553)
            scores, labels = torch.sigmoid(preds["cls_preds"].reshape(batch_size, -1, nc).float()).max(-1)
        k = min(cfg["nms_pre_max_size"], scores.shape[1])
        masked = torch.where(scores >= cfg["nms_score_threshold"], scores, torch.full_like(scores, -1.0))
        top_scores, top_idx = torch.topk(masked, k, dim=1)
        counts = (top_scores >= cfg["nms_score_threshold"]).sum(1).to(torch.int32)
        enc = torch.gather(box, 1, top_idx.unsqueeze(-1).expand(-1, -1, 7)).float()
        anc = anchors[top_idx] if anchors.dim() == 2 else torch.gather(anchors, 1, top_idx.unsqueeze(-1).expand(-1, -1, 7))
        dec = decode_boxes(enc, anc)
        if "dir_cls_preds" in preds:
            d = preds["dir_cls_preds"].reshape(batch_size, -1, cfg["num_direction_bins"])
            d = torch.gather(d, 1, top_idx.unsqueeze(-1).expand(-1, -1, cfg["num_direction_bins"]))
            dir_labels = torch.max(d, dim=-1)[1]
        else:
            dir_labels = None
        top_labels = torch.gather(labels, 1, top_idx) if labels is not None else torch.zeros_like(top_idx)
        return dec, top_scores, counts, dir_labels, top_labels

    def predict_device(self, preds, batch_size, anchors=None):
        """-> dict of padded device tensors: boxes [B,P,7], scores [B,P], labels [B,P], valid [B,P] (bool)."""
        cfg = self.cfg
        anchors = self.anchors if anchors is None else anchors
        if preds["cls_preds"].is_cuda and self.fused_predict and cfg["nms_pre_max_size"] <= 1024:
            # (sec_predict_select ranks up to 1024 candidates per frame; larger nms_pre_max_size takes the torch.topk path
            # below, which honours it -- never a silent truncation)
            return self._predict_fused(preds, batch_size, anchors)
        dec, top_scores, counts, dir_labels, top_labels = self._select(preds, batch_size, anchors)
        if cfg["use_rotate_nms"]:
            # (x, y, w, l, r, score): boxes_for_nms = box[:, [0, 1, 3, 4, 6]] (voxelnet.py:570); slices, not index
            # lists, so that nothing is staged from the host (hipGraph capture)
            dets = torch.cat([dec[..., 0:2], dec[..., 3:5], dec[..., 6:7], top_scores.unsqueeze(-1)], -1).contiguous()
            keep, num_keep = ops.nms_sorted(dets, counts, cfg["nms_iou_threshold"], "rotate", "cpu",
                                            post_max=cfg["nms_post_max_size"])
        else:
            # standup boxes of the rotated BEV rectangles (voxelnet.py:571-576 -> center_to_corner_box2d +
            # corner_to_standup_nd), then nms -> nms_gpu_cc -> spconv non_max_suppression ('+1', IoU > thr)
            hx, hy, ang = dec[..., 3] * 0.5, dec[..., 4] * 0.5, dec[..., 6]
            cs, sn = torch.cos(ang).abs(), torch.sin(ang).abs()
            ex, ey = hx * cs + hy * sn, hx * sn + hy * cs
            dets = torch.stack([dec[..., 0] - ex, dec[..., 1] - ey, dec[..., 0] + ex, dec[..., 1] + ey, top_scores], -1).contiguous()
            keep, num_keep = ops.nms_sorted(dets, counts, cfg["nms_iou_threshold"], "axis_aligned", "numba",
                                            post_max=cfg["nms_post_max_size"])
        p = min(cfg["nms_post_max_size"], keep.shape[1])
        valid = self._arange_p[:p].unsqueeze(0) < num_keep.unsqueeze(1)
        sel = torch.where(valid, keep[:, :p], torch.zeros_like(keep[:, :p])).long()   # slots past num_keep are undefined
        boxes = torch.gather(dec, 1, sel.unsqueeze(-1).expand(-1, -1, 7))
        scores = torch.gather(top_scores, 1, sel)
        if dir_labels is not None:
            dl = torch.gather(dir_labels, 1, sel)
            period = 2 * math.pi / cfg["num_direction_bins"]
            rot = limit_period(boxes[..., 6] - cfg["direction_offset"], cfg["direction_limit_offset"], period)
            boxes[..., 6] = rot + cfg["direction_offset"] + period * dl.to(boxes.dtype)
        r = self.post_center_range
        valid = valid & (boxes[..., :3] >= r[:3]).all(-1) & (boxes[..., :3] <= r[3:]).all(-1)
        return {"boxes": boxes, "scores": scores, "labels": torch.gather(top_labels, 1, sel), "valid": valid}

    def _predict_fused(self, preds, batch_size, anchors):
        """select -> decode -> NMS -> finalize: five launches, no host sync, no torch glue."""
        cfg = self.cfg
        a_per_loc = self.num_anchor_per_loc
        _, h, w = self.feature_map_size

        def view5(t, code):
            if t.dim() == 5:
                return t
            return t.reshape(batch_size, a_per_loc, h, w, code)
        cls = view5(preds["cls_preds"], cfg["num_class"])
        box = view5(preds["box_preds"], 7)
        dirp = view5(preds["dir_cls_preds"], cfg["num_direction_bins"]) if "dir_cls_preds" in preds else None
        if anchors.dim() == 3:
            anchors = anchors[0]
        anchors = anchors.float().contiguous()
        lz = preds.get("lazy_heads")       # (tile_live, empty-frame heads): background tiles of the head tensor were never written
        lz_cls = lz_box = None
        if lz is not None:
            bg = lz[1]
            lz_cls = (lz[0], view5(bg["cls_preds"], cfg["num_class"]))
            lz_box = (lz[0], (view5(bg["box_preds"], 7), view5(bg["dir_cls_preds"], cfg["num_direction_bins"]) if dirp is not None else None))
        top_idx, top_score, top_label, counts = ops.predict_select(cls, cfg["nms_pre_max_size"], cfg["nms_score_threshold"], lazy=lz_cls)
        dec, dets, dlab = ops.predict_decode(box, dirp, anchors, top_idx, top_score, rotate=cfg["use_rotate_nms"], lazy=lz_box)
        if cfg["use_rotate_nms"]:
            keep, num_keep = ops.nms_sorted(dets, counts, cfg["nms_iou_threshold"], "rotate", "cpu", post_max=cfg["nms_post_max_size"])
        else:
            keep, num_keep = ops.nms_sorted(dets, counts, cfg["nms_iou_threshold"], "axis_aligned", "numba",
                                            post_max=cfg["nms_post_max_size"])
        return ops.predict_finalize(dec, top_score, top_label, dlab, keep, num_keep, cfg["nms_post_max_size"],
                                    dirp is not None, cfg["direction_offset"], cfg["direction_limit_offset"],
                                    cfg["num_direction_bins"], self.post_center_range)

    def predict(self, preds, anchors):
        out = self.predict_device(preds, anchors.shape[0], anchors)
        res = []
        for b in range(anchors.shape[0]):
            m = out["valid"][b]
            res.append({"box3d_lidar": out["boxes"][b][m], "scores": out["scores"][b][m],
                        "label_preds": out["labels"][b][m], "metadata": None})
        return res


def lane_stream(device=None, index=None):
    """A stream for one lane of a serving loop.  HIP maps streams onto a handful of hardware queues (GPU_MAX_HW_QUEUES, default 4) per
    PRIORITY level, handing a new stream the least-shared queue: whether two lanes end up behind each other in one queue then depends
    on how many other streams the process created before (torch's capture / warm-up side streams included) -- measured on the same
    loop: 17.0 k, 13.3 k or 9.4 k frames/s depending on it.  High-priority streams come from a queue pool of their own that nothing
    else in the process uses, so a few lanes get a queue each whatever ran before.  SEC_LANE_PRIORITY=0: normal-priority streams."""
    import os
    groups = int(os.environ.get("SEC_LANE_CU_GROUPS", "0") or 0)
    if groups > 1 and index is not None:
        return _cu_masked_stream(index % groups, groups)
    prio = -1 if os.environ.get("SEC_LANE_PRIORITY", "-1") != "0" else 0
    return torch.cuda.Stream(device=device, priority=prio)


_masked_streams = []          # (hipStream_t, ExternalStream): kept alive for the life of the process


def _cu_masked_stream(group, groups):
    """EXPERIMENT (SEC_LANE_CU_GROUPS=g): a stream whose kernels run on the ``group``-th of ``groups`` equal slices of every XCD's CUs
    (hipExtStreamCreateWithCUMask; mask bit i = CU slot i // 8 of XCD i % 8 on MI355X, ``tools/probes/cu_mask_probe.hip``), so that lanes of
    different groups never share a CU and every chip-filling launch of a lane becomes several rounds on its own slice."""
    import ctypes
    path = next(l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l)
    hip = ctypes.CDLL(path)
    slots = 32 // groups                                   # CU slots per XCD and group
    bits = 0
    for slot in range(group * slots, (group + 1) * slots):
        bits |= 0xff << (8 * slot)
    words = (ctypes.c_uint32 * 8)(*[(bits >> (32 * w)) & 0xffffffff for w in range(8)])
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask -> {rc}")
    ext = torch.cuda.ExternalStream(st.value)
    _masked_streams.append((st, ext))
    return ext


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class InFlightRunner:
    """Serving loop with several steps in flight: ``inflight`` captured forwards (hipGraphs with their own activation
    buffers, all reading the same resident input buffers -- ``points`` / ``point_offsets``, or ``self.parts`` with branches)
    are replayed round-robin on their own HIP streams.  Replays on
    one lane serialise, different lanes overlap -- the latency-bound sparse stages of one step run beside the MFMA-bound
    RPN of another (car.fhd, batch 8: 5600 -> 7200 frames/s with three lanes; more lanes add nothing).
    ``serialize_rpn``: a step is three graphs (sparse front / RPN / predict) and the lanes hand a token (an event) from RPN
    segment to RPN segment, so that at most ONE lane is in its MFMA-bound segment at any time: uncoordinated lanes now and then
    sit in their RPN segments together (the others' launches wait for conv tiles to retire) or all in their latency-bound
    segments (matrix pipes idle).  Round 3: 14.0 k -> 14.7 k frames/s with four lanes (three lanes: 14.4 k; five: 13.3 k; the RPN
    segments on one extra stream with or without a CU mask instead of the token: 11.5 k -- the event ping-pong costs more than it
    orders; `gpurun_out` r03_as-au).

    ``step()`` enqueues one full forward and returns (outputs, stream): the output tensors of that lane, valid once
    ``stream`` has been synchronised (or after :meth:`synchronize`) and until the lane is stepped again."""

    def __init__(self, det, points, point_offsets, inflight=3, branches=1, private_inputs=False, serialize_rpn=False, rpn_tokens=None):
        """``private_inputs``: every lane gets its OWN copy of the input buffers (``self.inputs[k]``) and pinned host
        mirrors of its outputs, so that :meth:`step` can take a host-resident batch: the pinned-host -> HBM copy of lane k's
        next clouds then overlaps the compute of the other lanes (the end-to-end serving form; with shared buffers a copy
        would race with the lanes still reading them)."""
        self.det = det
        self.replays, self.outputs, self.parts = [], [], None
        self.inputs, self.host_outputs = [], []
        self._overflow = []                       # overflow counters of EVERY lane (each capture has its own rulebook buffers)
        # ``serialize_rpn``: every lane's step is three graphs (SecondDetector._capture_stages) and the RPN segments pass a token
        # (an event) from lane to lane: at most one lane is in its MFMA-bound segment at a time, the others' latency-bound
        # segments fill the rest of the chip
        self.serialize_rpn = bool(serialize_rpn) and branches <= 1 and int(inflight) > 1
        self._rpn_token = None
        # RPN segments allowed at a time (a ring of that many events); None = decided after the captures from the scene (below)
        self.rpn_tokens = None if rpn_tokens is None else max(1, int(rpn_tokens))
        self._rpn_events = []
        self._keepalive = []                      # per lane: the buffers its three graphs hand to each other
        assert not (private_inputs and branches > 1), "private input buffers are a single-chain feature"
        for _ in range(max(1, int(inflight))):
            if self.serialize_rpn:
                pk, ok = (points.clone(), point_offsets.clone()) if private_inputs else (points, point_offsets)
                replay, outs = det._capture_stages(pk, ok)
                self._keepalive.append(det._last_stage_buffers)
                self.inputs.append((pk, ok))
                if private_inputs:
                    self.host_outputs.append({k: torch.empty(v.shape, dtype=v.dtype, pin_memory=True) for k, v in outs.items()})
            elif branches > 1:
                # the per-branch input buffers are created by the first lane and shared by all others: one place to refill
                replay, outs, self.parts = det.make_graphed(points, point_offsets, branches=branches, parts=self.parts)
            else:
                pk, ok = (points.clone(), point_offsets.clone()) if private_inputs else (points, point_offsets)
                replay, outs = det.make_graphed(pk, ok)
                self.inputs.append((pk, ok))
                if private_inputs:
                    self.host_outputs.append({k: torch.empty(v.shape, dtype=v.dtype, pin_memory=True) for k, v in outs.items()})
            self.replays.append(replay)
            self.outputs.append(outs)
            self._overflow += [c for lst in getattr(det, "_branch_overflow", []) for c in lst]
        if self.rpn_tokens is None:
            # Two RPN segments at a time when the RPN is the light part of the step -- the calibration scene leaves most of the last conv's
            # tiles to the background -- and one when the scene is dense.  Measured with four lanes, interleaved runs (gpurun r06_lanes3 /
            # r06_tok2; round 6, since the last RPN conv carries the 1x1 tail): car.fhd (39 % of the tiles live) one token 20.2-20.3 k
            # frames/s, two 20.9 k, three 20.4-20.5 k, unserialised 20.5 k; dense seeded scene (81 % live) 8.45 k with one, 8.38 k with two.
            # (Before the tail moved, one and two tokens were level: 20.0 / 20.1 k, r05_j.)
            self.rpn_tokens = 1
            counts = getattr(det.rpn, "last_live_counts", None)
            if self.serialize_rpn and counts is not None and getattr(det.rpn, "skip_background", False):
                for seg in self.replays[-1]:          # the counts are the last capture's buffer: one replay of that lane fills it
                    seg()
                torch.cuda.synchronize()
                tiles = getattr(det.rpn, "last_tiles_per_frame", None)
                if tiles:
                    last = min(det.rpn.background_convs, counts.shape[0]) - 1
                    share = float(counts[last].sum().item()) / float(tiles * counts.shape[1])
                    self.rpn_tokens = 2 if share <= 0.6 else 1
        self.lanes = [lane_stream(index=i) for i in range(len(self.replays))] if len(self.replays) > 1 else [None]
        self._k = 0

    def step(self, host_points=None, host_offsets=None, fetch=False):
        """Enqueue one full forward on the next lane.  ``host_points`` [n <= capacity, F] / ``host_offsets`` [B + 1] (pinned
        host tensors; needs ``private_inputs``): copied into the lane's input buffers on the lane's stream ahead of the
        replay.  ``fetch``: the lane's detections are copied to its pinned host mirrors (``self.host_outputs[k]``) behind the
        replay.  Nothing here synchronises the host."""
        k = self._k % len(self.replays)
        self._k += 1
        lane = self.lanes[k]
        ctx = torch.cuda.stream(lane) if lane is not None else _NullCtx()
        with ctx:
            if host_points is not None:
                pk, ok = self.inputs[k]
                pk[:host_points.shape[0]].copy_(host_points, non_blocking=True)
                ok.copy_(host_offsets, non_blocking=True)
            if self.serialize_rpn:
                ra, rb, rc = self.replays[k]
                st = lane if lane is not None else torch.cuda.current_stream()
                ra()
                if len(self._rpn_events) >= self.rpn_tokens:
                    st.wait_event(self._rpn_events[-self.rpn_tokens])   # the RPN segment `rpn_tokens` steps back (another lane) has finished
                rb()
                ev = torch.cuda.Event()
                ev.record(st)
                self._rpn_events = (self._rpn_events + [ev])[-self.rpn_tokens:]
                rc()
            else:
                self.replays[k]()
            if fetch:
                for name, t in self.outputs[k].items():
                    self.host_outputs[k][name].copy_(t, non_blocking=True)
        return self.outputs[k], (lane if lane is not None else torch.cuda.current_stream())

    def synchronize(self):
        torch.cuda.synchronize()
        for num, cap in self._overflow:
            raw = int(num[1].item())
            if raw > cap:
                raise RuntimeError(f"static-capacity overflow: a strided sparse conv produced {raw} outputs, capacity {cap}")


def decode_boxes(enc, anc):
    """GroundBox3dCoder.decode (second_box_decode, second/pytorch/core/box_torch_ops.py:56-101)."""
    xa, ya, za, wa, la, ha, ra = anc.unbind(-1)
    xt, yt, zt, wt, lt, ht, rt = enc.unbind(-1)
    diag = torch.sqrt(la * la + wa * wa)
    return torch.stack([xt * diag + xa, yt * diag + ya, zt * ha + za, torch.exp(wt) * wa, torch.exp(lt) * la,
                        torch.exp(ht) * ha, rt + ra], -1)

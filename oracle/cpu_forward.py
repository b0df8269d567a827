"""CPU restatement of one whole VoxelNet.forward (car.fhd) for ONE frame -- TEST INFRASTRUCTURE, like everything in oracle/.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this; the product never does.
It follows the reference's stage order (second/pytorch/models/voxelnet.py:314-375 network, :377-645 predict):

    points_to_voxel          oracle C loop          (second/utils/simplevis.py:8-60; spconv point2voxel)
    SimpleVoxel              oracle                 (voxel_encoder.py:220-225)
    SpMiddleFHD              oracle rulebooks + indice_conv, BatchNorm1d(eval) + ReLU in numpy   (middle.py:145-210)
    RPNV2                    the topology mirror's torch modules on the CPU, fp32                 (rpn.py:468-497)
    predict                  sigmoid, >= score threshold, top-k, decode, rotated NMS (CPU semantics: standup pre-filter,
                             '>=' on the polygon IoU), direction fix, post-centre-range mask     (voxelnet.py:413-645)
"""
import math

import numpy as np
import torch

from . import oracle as orc


def limit_period_np(val, offset, period):
    return val - np.floor(val / period + offset) * period


def forward_frame(det, cloud, collect=False):
    """det: second_amd.models.SecondDetector on the CPU (fp32, eval).  cloud [N,4] float32.
    Returns dict(boxes [M,7], scores [M], labels [M], num_detections) and, with ``collect``, every intermediate the GPU
    pipeline can be compared against: voxel coordinates, every conv layer's output indices and features, the dense RPN
    input, the raw head outputs, the pre-NMS candidates and the NMS keep list."""
    from second_amd.models import decode_boxes
    cfg = det.cfg
    rec = {}
    v = orc.points_to_voxel(cloud, cfg["voxel_size"], cfg["point_cloud_range"], cfg["max_points_per_voxel"], cfg["max_voxels"])
    feat = orc.simple_voxel_mean(v["voxels"], v["num_points_per_voxel"], cfg["num_point_features"])
    idx = np.concatenate([np.zeros((v["voxel_num"], 1), np.int32), v["coordinates"]], 1)
    shape = [int(s) for s in det.middle_feature_extractor.sparse_shape]
    if collect:
        rec["voxel_coordinates"], rec["voxel_features"] = idx.copy(), feat.copy()
        rec["layers"] = []
    seq = list(det.middle_feature_extractor.middle_conv.children())
    cache = {}
    for i in range(0, len(seq), 3):
        conv, bn = seq[i], seq[i + 1]
        if conv.subm:
            if conv.indice_key not in cache:
                cache[conv.indice_key] = orc.rulebook_subm(idx, 1, shape, conv.kernel_size)
            out_idx, pairs, num = cache[conv.indice_key]
            n_out = len(idx)
        else:
            out_idx, pairs, num, oshape = orc.rulebook_conv(idx, 1, shape, conv.kernel_size, conv.stride, conv.padding)
            n_out = len(out_idx)
        y = orc.indice_conv(feat, conv.weight.detach().float().numpy(), pairs, num, n_out, acc64=False)
        scale = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).detach().float().numpy()
        shift = (bn.bias - bn.running_mean * torch.from_numpy(scale)).detach().float().numpy()
        feat = np.maximum(y * scale + shift, 0).astype(np.float32)
        if not conv.subm:
            idx, shape = out_idx, [int(s) for s in oshape]
        if collect:
            rec["layers"].append({"subm": bool(conv.subm), "out_indices": idx.copy(), "features": feat.copy(),
                                  "pairs": int(np.sum(num))})
    dense = orc.sparse_to_dense(feat, idx, 1, shape)
    x = torch.from_numpy(dense).view(1, -1, shape[1], shape[2])
    if collect:
        rec["spatial_features"] = x.numpy().copy()
    with torch.no_grad():
        preds = det.rpn(x.float())
        cls = torch.sigmoid(preds["cls_preds"].reshape(-1).float())
        keep = cls >= cfg["nms_score_threshold"]
        k = min(cfg["nms_pre_max_size"], int(keep.sum()))
        sel_all = torch.nonzero(keep).squeeze(1)
        # descending score, ties by ascending anchor index (the order the device top-k produces; torch.topk leaves ties
        # implementation-defined and the parity inputs have none that matter)
        order = torch.argsort(-cls[sel_all], stable=True)[:k]
        sel, sc = sel_all[order], cls[sel_all][order]
        boxes = decode_boxes(preds["box_preds"].reshape(-1, 7)[sel].float(), det.anchors[sel])
        dets = torch.cat([boxes[:, [0, 1, 3, 4, 6]], sc[:, None]], 1).numpy()
        dir_labels = torch.max(preds["dir_cls_preds"].reshape(-1, cfg["num_direction_bins"])[sel], -1)[1].numpy()
    kept = np.asarray(orc.rotate_nms_sorted(dets, cfg["nms_iou_threshold"], "cpu")[:cfg["nms_post_max_size"]], np.int64)
    out_boxes = boxes.numpy()[kept].copy()
    out_scores = sc.numpy()[kept]
    period = 2 * math.pi / cfg["num_direction_bins"]
    rot = limit_period_np(out_boxes[:, 6] - np.float32(cfg["direction_offset"]), np.float32(cfg["direction_limit_offset"]),
                          np.float32(period))
    out_boxes[:, 6] = rot + np.float32(cfg["direction_offset"]) + np.float32(period) * dir_labels[kept].astype(np.float32)
    r = np.asarray(cfg["post_center_range"], np.float32)
    m = (out_boxes[:, :3] >= r[:3]).all(1) & (out_boxes[:, :3] <= r[3:]).all(1)
    res = {"boxes": out_boxes[m], "scores": out_scores[m], "labels": np.zeros(int(m.sum()), np.int32),
           "num_detections": int(m.sum())}
    if collect:
        rec.update(cls_preds=preds["cls_preds"].numpy(), box_preds=preds["box_preds"].numpy(),
                   dir_cls_preds=preds["dir_cls_preds"].numpy(), candidate_anchor_ids=sel.numpy(), candidate_scores=sc.numpy(),
                   candidate_boxes=boxes.numpy(), nms_keep=kept, range_mask=m)
        res["trace"] = rec
    return res

"""Per-stage comparison of the DEVICE forward (bf16 / fp16 / fp32, static capacities, the kernels bench.py times) with the CPU
restatement of VoxelNet.forward (oracle/cpu_forward.py) -- test infrastructure shared by tests/test_gpu_e2e.py and
tools/bf16_error_trace.py (second/pytorch/models/voxelnet.py:314-375 network, :377-645 predict).

Two error figures per layer, because they answer different questions:

  cumulative   max |device - cpu_fp32| / max |cpu_fp32| of the layer's output, the CPU chain running in fp32 from the raw points:
               how far 16-bit storage of the activations has drifted after this many layers (sparse rows are aligned through their
               coordinates, which must agree exactly -- rulebook / numbering parity at bench size comes for free);
  single       the layer re-computed on the CPU from the DEVICE's own 16-bit input, 16-bit weights and fp32 scale / shift, compared
               with the device's 16-bit output in units of the only error that output may carry: the unit roundoff of the storage
               type (round to nearest: 2^-8 relative for bf16's 8 significant bits, 2^-11 for fp16) plus BASELINE.json's 1e-4 of
               the layer's range.  <= ~1 means the kernel's arithmetic is exact up to the rounding of
               its stored result; it isolates a wrong kernel from accumulated rounding."""
import numpy as np
import torch

from oracle import oracle as orc
from oracle.cpu_forward import forward_frame


def trained_like_detector(cfg, calib_cloud, seed=0):
    """CPU fp32 detector with the synthetic 'trained-like' weights bench.py uses (second_amd.synthetic), heads calibrated on
    ``calib_cloud`` through the CPU oracle forward."""
    from second_amd import synthetic as syn
    from second_amd.models import SecondDetector
    torch.manual_seed(seed)
    det = SecondDetector(cfg).eval()
    syn.randomise_like_trained(det, seed=1)
    tr = forward_frame(det, calib_cloud, collect=True)["trace"]
    syn.sharpen_heads(det, tr["cls_preds"], tr["box_preds"])
    return det


def _lin(idx, shape):
    return (idx[:, 1].astype(np.int64) * shape[1] + idx[:, 2]) * shape[2] + idx[:, 3]


def _pairs_from_table(nbr):
    """spconv pair lists [K, 2, N_in-capacity] / counts [K] from an output-major gather table nbr[o][k] = input row or -1."""
    n_out, k = nbr.shape
    cap = max(int(nbr.max()) + 1, n_out, 1)
    pairs = np.full((k, 2, cap), -1, np.int32)
    num = np.zeros(k, np.int32)
    for kk in range(k):
        o = np.nonzero(nbr[:, kk] >= 0)[0]
        num[kk] = len(o)
        pairs[kk, 0, :len(o)] = nbr[o, kk]
        pairs[kk, 1, :len(o)] = o
    return pairs, num, cap


def _unit_err(got, ref, ulp):
    """max |got - ref| in units of (unit roundoff of the 16-bit result, `ulp` * |ref|, + 1e-4 of the layer's range)."""
    tol = ulp * np.abs(ref) + 1e-4 * max(float(np.abs(ref).max()), 1e-30)
    return float((np.abs(got - ref) / tol).max())


def run_device_trace(gpu, points, offsets):
    """One static-capacity EAGER forward of ``gpu`` (SecondDetector on the device, any dtype) with every traced op recorded:
    the same kernels with the same launch arguments a captured graph replays."""
    from second_amd import ops
    calls = []
    ops.set_op_hook(lambda name, fn, a, kw, res: calls.append((name, a, kw, res)))
    # the trace compares WHOLE layer outputs: the RPN convs materialise their background tiles here (lazy_background off; the live
    # tiles -- all the arithmetic there is -- are bit-identical either way, tests/test_gpu_rpn_tiles.py)
    lazy = getattr(gpu.rpn, "lazy_background", None)
    if lazy:
        gpu.rpn.lazy_background = False
    try:
        with torch.no_grad():
            out = gpu.forward_points(points, offsets, static=True)
        torch.cuda.synchronize()
    finally:
        ops.set_op_hook(None)
        if lazy:
            gpu.rpn.lazy_background = lazy
    return calls, out


def sparse_stage_errors(calls, refs, ulp, sparse_shape, single=True):
    """-> list of dicts per sparse conv layer.  ``refs``: forward_frame(..., collect=True)["trace"] per frame;
    ``sparse_shape``: the middle extractor's input grid (z, y, x)."""
    nb = len(refs)
    vox = [c for c in calls if c[0] == "voxelize"][0][3]
    n_live = int(vox["voxel_offsets"][nb].item())
    idx = vox["coordinates"][:n_live].cpu().numpy()
    shape = [int(v) for v in sparse_shape]
    rows = []
    li = 0
    # voxel coordinates: bit-exact per frame, in the reference's first-occurrence order (simplevis.py:31-50)
    voffs = vox["voxel_offsets"].cpu().numpy()
    for f, r in enumerate(refs):
        got = idx[voffs[f]:voffs[f + 1]]
        assert np.array_equal(got[:, 1:], r["voxel_coordinates"][:, 1:]) and np.all(got[:, 0] == f), f"voxel coordinates of frame {f}"
    by_table = {}          # fused rulebook chain: gather table (data_ptr) -> (output sites, grid) of the level it belongs to
    for name, a, kw, res in calls:
        if name == "rulebook_conv":
            m = int(res["num_out_dev"][0].item()) if res.get("num_out_dev") is not None else int(res["num_out"])
            idx, shape = res["out_indices"][:m].cpu().numpy(), [int(s) for s in res["out_shape"]]
        if name == "rulebook_chain" and res is not None:
            for L in res["levels"]:
                m = int(L["num_dev"].reshape(-1)[0].item()) if L["num_dev"] is not None else int(L["cap"])
                ent = (L["indices"][:m].cpu().numpy(), [int(v) for v in L["shape"]])
                for t in (L["nbr_out"], L["subm_nbr"]):
                    if t is not None:
                        by_table[t.data_ptr()] = ent
        if name != "indice_conv":
            continue
        feat, w, nbr, cap = a[:4]
        if nbr.data_ptr() in by_table:
            idx, shape = by_table[nbr.data_ptr()]
        m = int(kw["num_out_dev"][0].item()) if kw.get("num_out_dev") is not None else int(cap)
        got = res[:m].float().cpu().numpy()
        assert idx.shape[0] == m, (li, idx.shape, m)
        lshape = shape
        ent = {"layer": li, "kind": "SubM" if refs[0]["layers"][li]["subm"] else "strided", "cin": int(w.shape[-2]), "cout": int(w.shape[-1]),
               "rows": m, "cumulative": 0.0, "range": 0.0}
        for f, r in enumerate(refs):
            l = r["layers"][li]
            sel = idx[:, 0] == f
            kg, kc = _lin(idx[sel], lshape), _lin(l["out_indices"], lshape)
            og, oc = np.argsort(kg, kind="stable"), np.argsort(kc, kind="stable")
            assert np.array_equal(kg[og], kc[oc]), f"layer {li} frame {f}: active sites differ from the CPU rulebook"
            ref = l["features"][oc]
            ent["cumulative"] = max(ent["cumulative"], float(np.abs(got[sel][og] - ref).max() / max(np.abs(ref).max(), 1e-30)))
            ent["range"] = max(ent["range"], float(np.abs(ref).max()))
        if single:
            # single-layer: the oracle's indice_conv on the DEVICE's input rows through the DEVICE's gather table (the rulebooks
            # themselves are bit-exact, asserted above through the sites and in tests/test_gpu_parity.py through the pairs)
            pairs, num, capp = _pairs_from_table(nbr[:m].cpu().numpy())
            fin = np.zeros((capp, feat.shape[1]), np.float32)
            src = feat.float().cpu().numpy()
            fin[:min(capp, src.shape[0])] = src[:capp]
            y = orc.indice_conv(fin, w.float().cpu().numpy(), pairs, num, m, acc64=False)
            sc, sh = kw["scale"].cpu().numpy(), kw["shift"].cpu().numpy()
            y = y * sc + sh
            if kw.get("relu"):
                y = np.maximum(y, 0)
            ent["single"] = _unit_err(got, y.astype(np.float32), ulp)
        rows.append(ent)
        li += 1
    return rows


def dense_stage_errors(calls, gpu, det_cpu, refs, ulp, single_frames=2):
    """RPN: cumulative error of every 3x3 layer's output and of the heads vs the fp32 CPU chain; single-layer error of each layer
    (torch CPU conv2d on the device's own input with the device's folded 16-bit weights) on the first ``single_frames`` frames."""
    import torch.nn.functional as F
    nb = len(refs)
    rows = []
    blk = list(det_cpu.rpn.blocks[0].children())
    rpn = gpu.rpn
    convs = [c for c in calls if c[0] in ("conv2d_nhwc_gather", "conv2d_nhwc", "conv2d_nhwc_tiles")]
    cum, rng = [0.0] * len(convs), [0.0] * len(convs)
    with torch.no_grad():
        for f, r in enumerate(refs):            # fp32 CPU activation after every Conv + BN + ReLU of the block, frame by frame
            x = torch.from_numpy(r["spatial_features"]).float()
            i = 0
            for m in blk:
                x = m(x)
                if isinstance(m, torch.nn.ReLU):
                    ref = x[0].numpy()
                    got = convs[i][3][f].float().cpu().numpy()
                    cum[i] = max(cum[i], float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30)))
                    rng[i] = max(rng[i], float(np.abs(ref).max()))
                    i += 1
            assert i == len(convs), (i, len(convs))
    prev = None
    for i, (name, a, kw, res) in enumerate(convs):
        ent = {"layer": f"rpn{i}", "kind": "3x3 gathered from sparse rows" if name == "conv2d_nhwc_gather" else "3x3 (live tiles)" if name == "conv2d_nhwc_tiles" else "3x3", "cin": 128,
               "cout": int(res.shape[1]), "cumulative": cum[i], "range": rng[i]}
        if single_frames:
            got = res[:single_frames].float().cpu().numpy()
            if name == "conv2d_nhwc_gather":
                # the device's dense image, rebuilt on the host from the rows the gather reads: channel = c * D + z
                feat, smap = a[0].float().cpu().numpy(), a[1].cpu().numpy()
                xin = np.zeros((single_frames, 128, smap.shape[2], smap.shape[3]), np.float32)
                for f in range(single_frames):
                    for z in range(smap.shape[1]):
                        yy, xx = np.nonzero(smap[f, z])
                        xin[f, z::2, yy, xx] = feat[smap[f, z, yy, xx] - 1]
                xin = torch.from_numpy(xin)
            else:
                xin = prev[:single_frames].float().cpu()
            with torch.no_grad():
                (s, p) = rpn.cfgs[i]
                ref = F.relu(F.conv2d(xin, rpn.ws[i].float().cpu(), rpn.bs[i].float().cpu(), s, p)).numpy()
            ent["single"] = _unit_err(got, ref, ulp)
        prev = res
        rows.append(ent)
    chain = [c for c in calls if c[0] == "conv1x1_chain"]
    if chain:
        name, a, kw, res = chain[0]
        got = res.float().cpu().numpy()
        refs = [r if "box_preds" in r else r["trace"] for r in refs]
        ent = {"layer": "rpn-tail", "kind": "1x1 deblock + 1x1 heads (fused)", "cin": 128, "cout": int(got.shape[1])}
        # cumulative: the three head tensors against the CPU heads
        worst, rng = 0.0, 0.0
        c0 = 0
        for key, sz in zip(["box_preds", "cls_preds", "dir_cls_preds"], rpn.splits):
            for f, r in enumerate(refs):
                ref = r[key][0]                                  # [A, H, W, code]
                g = got[f, c0:c0 + sz].reshape(ref.shape[0], ref.shape[3], ref.shape[1], ref.shape[2]).transpose(0, 2, 3, 1)
                worst = max(worst, float(np.abs(g - ref).max() / np.abs(ref).max()))
                rng = max(rng, float(np.abs(ref).max()))
            c0 += sz
        ent["cumulative"], ent["range"] = worst, rng
        if single_frames:
            xin = a[0][:single_frames].float().cpu()
            with torch.no_grad():
                (s, p) = rpn.cfgs[-1]
                mid = F.relu(F.conv2d(xin, rpn.ws[-1].float().cpu(), rpn.bs[-1].float().cpu(), s, p))
                mid = mid.to(res.dtype).float()                 # the 128-channel intermediate is stored 16-bit (in LDS)
                hw = torch.cat([rpn.head_w.float().cpu(), torch.zeros(got.shape[1] - rpn.head_w.shape[0], 128, 1, 1)], 0)
                hb = rpn.head_b64.float().cpu()
                ref = F.conv2d(mid, hw, hb).numpy()
            # two stacked roundings: an ulp flip of the intermediate moves the result by <= 2^-8 |mid| |w|; allow 2 units
            ent["single"] = _unit_err(got[:single_frames], ref, ulp) / 2.0
        rows.append(ent)
    return rows


def match_detections(out, refs, dist=0.25, dscore=0.05):
    """CPU detections (``refs``: forward_frame results) found on the device (same frame, BEV centre within ``dist``, score
    within ``dscore``).  -> found, total, [(count_device, count_cpu)], [(frame, cpu detection index) not found]."""
    gb, gs, gv = out["boxes"].float().cpu().numpy(), out["scores"].float().cpu().numpy(), out["valid"].cpu().numpy().astype(bool)
    found, total, counts, missed = 0, 0, [], []
    for f, r in enumerate(refs):
        m = gv[f]
        counts.append((int(m.sum()), int(r["num_detections"])))
        for j, (bx, sc) in enumerate(zip(r["boxes"], r["scores"])):
            total += 1
            ok = False
            if m.any():
                d = np.hypot(gb[f][m][:, 0] - bx[0], gb[f][m][:, 1] - bx[1])
                ok = bool(((d < dist) & (np.abs(gs[f][m] - sc) < dscore)).any())
            found += ok
            if not ok:
                missed.append((f, j))
    return found, total, counts, missed


def attribute_misses(calls, refs, missed, score_thr):
    """Why a CPU detection is absent on the device, from the device's own selection / NMS records:
      below_threshold   its anchor's device score fell below nms_score_threshold (or out of the top-k): a score difference;
      suppressed        its anchor is a device candidate but the device's NMS dropped it: an overlapping neighbour outranks it
                        on the device (scores that differ by less than the 16-bit drift swap places; iou_threshold 0.01 makes
                        any overlap decisive);
      drifted           kept on the device, but its score or centre moved by more than the matching tolerance."""
    sel = [c for c in calls if c[0] == "predict_select"][0][3]
    nms = [c for c in calls if c[0] == "nms_sorted"][0][3]
    top_idx, counts = sel[0].cpu().numpy(), sel[3].cpu().numpy()
    keep, num_keep = nms[0].cpu().numpy(), nms[1].cpu().numpy()
    why = {"below_threshold": 0, "suppressed": 0, "drifted": 0}
    for f, j in missed:
        tr = refs[f]["trace"]
        kept_cpu = tr["nms_keep"][tr["range_mask"]] if "range_mask" in tr else tr["nms_keep"]
        anchor = int(tr["candidate_anchor_ids"][kept_cpu[j]])
        cand = top_idx[f, :counts[f]]
        pos = np.nonzero(cand == anchor)[0]
        if len(pos) == 0:
            why["below_threshold"] += 1
        elif int(pos[0]) not in set(keep[f, :num_keep[f]].tolist()):
            why["suppressed"] += 1
        else:
            why["drifted"] += 1
    return why


def format_table(rows):
    out = [f"{'layer':>9s} {'kind':34s} {'shape':>9s} {'rows':>7s} {'cumulative (rel to range)':>26s} {'single (units of roundoff + 1e-4)':>33s}"]
    for r in rows:
        out.append(f"{str(r['layer']):>9s} {r['kind']:34s} {str(r['cin']) + '->' + str(r['cout']):>9s} {str(r.get('rows', '')):>7s} "
                   f"{r['cumulative']:26.5f} {r.get('single', float('nan')):33.3f}")
    return "\n".join(out)

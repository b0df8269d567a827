"""Pin the CPU oracle against fixtures produced by executing the reference's own Python
(tests/golden/make_golden.py).  CPU-only."""
import numpy as np
import pytest

from oracle import oracle as orc


@pytest.mark.parametrize("tag", ["kitti", "coarse_cap"])
def test_voxel_coordinates_match_simplevis(golden, tag):
    """second/utils/simplevis.py:8-60 (in-repo copy of spconv's points_to_voxel loop):
    voxel (z,y,x) coordinates, first-occurrence numbering and `break` at the cap."""
    g = golden("voxel_coords")
    pts, vs, rng, cap = g[f"{tag}_points"], g[f"{tag}_voxel_size"], g[f"{tag}_range"], int(g[f"{tag}_cap"])
    res = orc.points_to_voxel(pts, vs, rng, max_points=5, max_voxels=cap, cap_mode="break")
    assert res["voxel_num"] == len(g[f"{tag}_coors"])
    np.testing.assert_array_equal(res["coordinates"], g[f"{tag}_coors"])
    if tag == "coarse_cap":
        assert res["voxel_num"] == cap
        # `continue` mode keeps filling existing voxels after the cap: same voxels, >= points
        res_c = orc.points_to_voxel(pts, vs, rng, 5, cap, cap_mode="continue")
        np.testing.assert_array_equal(res_c["coordinates"], res["coordinates"])
        assert (res_c["num_points_per_voxel"] >= res["num_points_per_voxel"]).all()
        assert res_c["num_points_per_voxel"].sum() > res["num_points_per_voxel"].sum()


def test_grid_size(golden):
    np.testing.assert_array_equal(orc.grid_size([0, -40, -3, 70.4, 40, 1], [0.05, 0.05, 0.1]), [1408, 1600, 40])
    np.testing.assert_array_equal(orc.grid_size([-50, -50, -5, 50, 50, 3], [0.25, 0.25, 8]), [400, 400, 1])


def test_standup_prefilter(golden):
    """box_np_ops.center_to_corner_box2d + corner_to_standup_nd + iou_jit(eps=0) (nms_cpu.py:17-28)."""
    g = golden("standup")
    sb, iou = orc.standup_iou(g["dets"])
    np.testing.assert_allclose(sb, g["standup"], atol=2e-6, rtol=0)
    np.testing.assert_allclose(iou, g["standup_iou"], atol=2e-6, rtol=0)
    assert ((iou > 0) == (g["standup_iou"] > 0)).all()


def test_simple_voxel_and_decode(golden):
    g = golden("torch_modules")
    out = orc.simple_voxel_mean(g["sv_voxels"], g["sv_num_points"], 4)
    np.testing.assert_allclose(out, g["sv_out"], rtol=1e-6, atol=1e-7)
    dec = orc.box_decode(g["dec_enc"], g["dec_anchors"])
    np.testing.assert_allclose(dec, g["dec_out"], rtol=2e-6, atol=1e-6)


def test_pillar_scatter(golden):
    g = golden("torch_modules")
    b, c, ny, nx = g["ps_out"].shape
    out = orc.pillar_scatter(g["ps_feats"], g["ps_coords"], b, ny, nx)
    np.testing.assert_array_equal(out, g["ps_out"])


def test_rotate_iou_matches_numba_spec(golden):
    """nms_gpu.py rotate_iou_gpu_eval (:564-640) for all four criteria."""
    g = golden("rotate_iou")
    for crit in (-1, 0, 1, 2):
        out = orc.rotate_iou(g["boxes"], g["qboxes"], crit)
        np.testing.assert_allclose(out, g[f"iou_c{crit}"], atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(orc.rotate_iou(g["boxes"], g["qboxes"]), g["iou_plain"], atol=2e-5)
    sp = np.array([orc.rotate_iou(g["special_a"][i:i + 1], g["special_b"][i:i + 1])[0, 0]
                   for i in range(len(g["special_a"]))])
    np.testing.assert_allclose(sp, g["special_iou"], atol=2e-5)
    # hand-computed known answers: identical squares, half shift, 45 degree turn
    np.testing.assert_allclose(sp[:3], [1.0, 1.0 / 3.0, 1.0 / np.sqrt(2.0)], atol=1e-5)


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
@pytest.mark.parametrize("thr", [0.01, 0.3])
def test_rotate_nms_matches_numba_spec(golden, tag, thr):
    """nms_gpu.py rotate_nms_gpu (:440-475): keep list in original indices."""
    g = golden("rotate_nms")
    dets = g[f"dets_{tag}"]
    order = dets[:, 5].argsort()[::-1]
    keep = order[orc.rotate_nms_sorted(dets[order], thr, "numba")]
    np.testing.assert_array_equal(keep, g[f"keep_{tag}_{thr}"])
    # CPU semantics (standup pre-filter, >=) agree away from threshold ties
    keep_cpu = order[orc.rotate_nms_sorted(dets[order], thr, "cpu")]
    np.testing.assert_array_equal(keep_cpu, g[f"keep_{tag}_{thr}"])


@pytest.mark.parametrize("thr", [0.1, 0.5])
def test_axis_aligned_nms(golden, thr):
    """nms_gpu.nms_gpu ('+1', '>') and nms_cpu.nms_jit (eps, '>=')."""
    g = golden("nms_axis_aligned")
    dets = g["dets"]
    order = dets[:, 4].argsort()[::-1]
    np.testing.assert_array_equal(order[orc.nms_sorted(dets[order], thr, "numba")], g[f"keep_gpu_{thr}"])
    np.testing.assert_array_equal(order[orc.nms_sorted(dets[order], thr, "cpu", 0.0)], g[f"keep_jit_eps0_{thr}"])
    np.testing.assert_array_equal(order[orc.nms_sorted(dets[order], thr, "cpu", 1.0)], g[f"keep_jit_eps1_{thr}"])

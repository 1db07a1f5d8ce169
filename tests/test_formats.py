"""Row f4: on-disk formats (COLMAP binary model, LLFF poses_bounds, sparse depth supervision, DPT prior, checkpoint pair)
against a tiny model written by the reference's own COLMAP writer and parsed / processed by the reference's own readers
and dataset code (tests/golden/make_golden.py g14 -> tests/golden/colmap_tiny/, g14_formats.npz).  CPU only."""
import os

import numpy as np
import torch

from conftest import load_golden
from uc_nerf_amd.data import formats as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "golden", "colmap_tiny")


def g14():
    return {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in load_golden("g14_formats").items()}


def test_colmap_binary_readers_match_the_reference_readers():
    g = g14()
    cams = F.read_cameras_binary(os.path.join(ROOT, "sparse", "0", "cameras.bin"))
    assert list(cams) == [1] and cams[1].model == "PINHOLE"
    assert np.array_equal(cams[1].params, g["cam_params"]) and [cams[1].width, cams[1].height] == g["cam_wh"].tolist()
    imgs = F.read_images_binary(os.path.join(ROOT, "sparse", "0", "images.bin"))
    assert [imgs[i].name for i in imgs] == [str(n) for n in g["img_names"]]
    for i in imgs:
        assert np.array_equal(imgs[i].qvec, g["img%d_qvec" % i]) and np.array_equal(imgs[i].tvec, g["img%d_tvec" % i])
        assert np.array_equal(imgs[i].xys, g["img%d_xys" % i]) and np.array_equal(imgs[i].point3D_ids, g["img%d_ids" % i])
        np.testing.assert_allclose(imgs[i].qvec2rotmat(), g["img%d_R" % i], rtol=0, atol=1e-15)
    pts = F.read_points3d_binary(os.path.join(ROOT, "sparse", "0", "points3D.bin"))
    ids = sorted(pts)
    assert ids == g["pt_ids"].tolist()
    assert np.array_equal(np.stack([pts[k].xyz for k in ids]), g["pt_xyz"])
    assert np.array_equal(np.array([float(pts[k].error) for k in ids]), g["pt_err"])
    assert np.array_equal(np.stack([pts[k].rgb for k in ids]), g["pt_rgb"])
    assert np.array_equal(np.stack([pts[7].image_ids, pts[7].point2D_idxs]), g["pt7_track"])


def test_llff_pose_normalisation_and_source_view_selection():
    g = g14()
    rec = F.load_poses_bounds(os.path.join(ROOT, "poses_bounds.npy"), img_wh=(32, 24))
    np.testing.assert_allclose(rec["poses"], g["poses_centred"], rtol=0, atol=1e-14)
    np.testing.assert_allclose(rec["pose_avg"], g["pose_avg"], rtol=0, atol=1e-14)
    assert np.array_equal(rec["bounds"], g["bounds_scaled"]) and rec["scale_factor"] == float(g["scale_factor"])
    assert rec["focal"] == [50.0 * 32 / 64, 50.0 * 24 / 48]
    # centring: the average of the centred poses is the flipped identity frame
    avg = F.average_poses(rec["poses"])
    np.testing.assert_allclose(avg[:, 3], 0, atol=1e-12)
    near = F.get_nearest_pose_ids(rec["poses"][2], rec["poses"][[0, 1, 3, 4]], 3)
    assert near.tolist() == g["nearest"].tolist()


def test_sparse_depth_supervision_matches_the_dataset_code():
    g = g14()
    imgs = F.read_images_binary(os.path.join(ROOT, "sparse", "0", "images.bin"))
    pts = F.read_points3d_binary(os.path.join(ROOT, "sparse", "0", "points3D.bin"))
    pb = np.load(os.path.join(ROOT, "poses_bounds.npy"))
    lst = F.colmap_sparse_depth(imgs, pts, pb[:, -2:].transpose([1, 0]), img_wh=(32, 24), factor=2.0)
    assert [d["name"] for d in lst] == [str(n) for n in g["sd_names"]] == sorted(d["name"] for d in lst)
    for k, d in enumerate(lst):
        np.testing.assert_allclose(d["depth"], g["sd%d_depth" % k], rtol=1e-14, atol=0)
        assert np.array_equal(d["coord"], g["sd%d_coord" % k])
        np.testing.assert_allclose(d["weight"], g["sd%d_weight" % k], rtol=1e-14, atol=0)
        np.testing.assert_allclose(d["depth_img"], g["sd%d_depth_img" % k], rtol=1e-14, atol=0)
        np.testing.assert_allclose(d["weight_img"], g["sd%d_weight_img" % k], rtol=1e-14, atol=0)


def test_dpt_prior_and_checkpoint_pair_round_trip(tmp_path):
    pred = np.random.RandomState(0).rand(1, 12, 16).astype(np.float32)
    np.savez(tmp_path / "frame.npz", pred=pred)
    assert np.array_equal(F.load_dpt_prior(tmp_path / "frame.npz"), pred[0])
    # the trainer's pair of state dicts (train.py:403-413) with the reference's parameter names
    from uc_nerf_amd.synthetic import init_ucnerf_state_dict
    sd = init_ucnerf_state_dict(seed=3, n_src=6)
    mvs = {"feature.conv0.0.conv.weight": torch.randn(8, 3, 3, 3)}
    F.save_checkpoint(str(tmp_path / "ckpts" / "latest.tar"), sd, mvs)
    raw = torch.load(tmp_path / "ckpts" / "latest.tar", map_location="cpu")
    assert set(raw) == {"network_fn_state_dict", "network_mvs_state_dict"}
    back = F.load_checkpoint(str(tmp_path / "ckpts" / "latest.tar"))
    assert all(torch.equal(back["network_fn_state_dict"][k], v) for k, v in sd.items())
    # ... and into a module with the reference's layout (view_num 7 -> feat_dim 97)
    from uc_nerf_amd.network.models import UCNeRF
    net = UCNeRF(D=6, W=128, input_ch_pts=63, input_ch_views=27, input_ch_feat=97, view_num=7)
    assert set(net.state_dict()) == set(sd)
    F.load_checkpoint(str(tmp_path / "ckpts" / "latest.tar"), network_fn=net)
    assert all(torch.equal(net.state_dict()[k].cpu(), v) for k, v in sd.items())

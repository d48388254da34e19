"""Caller-side rows next to the rasterizer against fixtures G12, which tests/golden/
make_goldens.py --handoff-only produced by running the reference's own definitions (HFTSConfig,
create_camera_from_pose, ImageDataset, rotate_positions_for_pose), plus format known-answers for the C++ viewer's
PLY (src/core/renderer/renderer.cpp:649-793, unbuildable here: written from the format it defines)."""
import json
import os

import numpy as np
import pytest
import torch

from helpers import load_golden


@pytest.fixture(scope="module")
def g12():
    return load_golden("G12_handoff")


def test_hfts_config_tables_match_reference(g12):
    from fresnel_amd.handoff import HFTSConfig
    for ci, kw in enumerate(g12["hfts_configs"]):
        h = HFTSConfig(**json.loads(str(kw)))
        for ti, (T, b) in enumerate([(20, 4), (20, 8), (7, 2), (0, 4)]):
            got = [h.get_gaussians_per_patch(e, T, b) for e in range(0, 21)]
            assert got == g12["hfts_gpp"][ci][ti].tolist(), (kw, T, b)
        got_k = [-1 if h.get_stochastic_k(n) is None else h.get_stochastic_k(n) for n in (100, 256, 5476)]
        assert got_k == g12["hfts_k"][ci].tolist()
        assert [h.get_effective_train_resolution(s) for s in (64, 256, 512)] == g12["hfts_res"][ci].tolist()


def test_create_camera_from_pose_matches_reference(g12):
    from fresnel_amd.renderer import create_camera_from_pose
    for i, (el, az) in enumerate(g12["pose_deg"]):
        cam = create_camera_from_pose(np.radians(el), np.radians(az), 96)
        assert np.allclose(cam.view_matrix.numpy(), g12["pose_view"][i], atol=1e-6), (el, az)
        assert np.allclose([cam.fx, cam.fy, cam.cx, cam.cy, cam.width, cam.height, cam.near, cam.far], g12["pose_intr"][i])
    cam = create_camera_from_pose(0.3, 1.1, 128, focal_length_mult=1.2, distance=3.5)
    assert np.allclose(cam.view_matrix.numpy(), g12["pose_view"][-1], atol=1e-6)
    assert np.allclose([cam.fx, cam.fy, cam.cx, cam.cy], g12["pose_intr"][-1][:4])


def test_rotate_positions_for_pose_matches_reference(g12):
    from fresnel_amd.decoder import rotate_positions_for_pose
    out = rotate_positions_for_pose(torch.from_numpy(g12["rot_in"]), torch.from_numpy(g12["rot_el"]), torch.from_numpy(g12["rot_az"]))
    assert np.abs(out.numpy() - g12["rot_out"]).max() <= 1e-6


def test_image_dataset_matches_reference(g12, tmp_path):
    """The data files of the fixture are written out again and read by fresnel_amd.data.ImageDataset: image (PNG +
    LANCZOS), patch-major feature cache, depth cache at native size and through the 8-bit bilinear resize, SAAG
    binary, missing-cache fall-backs, sorted discovery, max_images, the 768-dim cache suffix."""
    from fresnel_amd.data import ImageDataset
    for rel in g12["ds_files"]:
        full = tmp_path / str(rel)
        full.parent.mkdir(parents=True, exist_ok=True)
        full.write_bytes(g12["file:" + str(rel)].tobytes())
    S, FD = int(g12["ds_image_size"]), int(g12["ds_feature_dim"])
    ds = ImageDataset(str(tmp_path), image_size=S, feature_dim=FD)
    assert len(ds) == 3
    for i in range(3):
        it = ds[i]
        assert it["name"] == str(g12[f"ds{i}_name"]) and bool(it["has_saag"]) == bool(g12[f"ds{i}_has_saag"])
        assert np.abs(it["image"].numpy() - g12[f"ds{i}_image"]).max() <= 1.0 / 255 + 1e-6
        assert np.array_equal(it["features"].numpy(), g12[f"ds{i}_features"])
        assert np.abs(it["depth"].numpy() - g12[f"ds{i}_depth"]).max() <= 1.0 / 255 + 1e-6
        for k in ("positions", "scales", "rotations", "colors", "opacities"):
            assert np.array_equal(it["saag_" + k].numpy(), g12[f"ds{i}_saag_{k}"]), k
    ds2 = ImageDataset(str(tmp_path), image_size=S, max_images=2, feature_dim=768)
    assert len(ds2) == int(g12["ds_max2_len"]) and ds2.feature_suffix == str(g12["ds_suffix_768"])
    images, feats, depth = ds.batch([0, 2], "cpu")
    assert images.shape == (2, 3, S, S) and feats.shape == (2, 37, 37, FD) and depth.shape == (2, 1, S, S)
    assert torch.equal(feats[0], ds[0]["features"].permute(1, 2, 0))


def test_importance_weights_and_pose_sampling():
    from fresnel_amd.handoff import importance_weights, sample_training_pose
    opa = torch.tensor([[0.0, 0.5, 1.0], [0.2, 0.5, 0.0]])
    w = importance_weights(opa)
    ref = opa.mean(0) + 1e-6
    assert torch.allclose(w, ref / ref.sum()) and abs(float(w.sum()) - 1.0) < 1e-6 and float(w.min()) > 0
    assert sample_training_pose(False, True, 0.3) == (None, None, True)
    assert sample_training_pose(True, False, 0.3) == (None, None, True)   # needs --use_pose_encoding too (TGD:1082)
    rng, ref_rng = np.random.RandomState(5), np.random.RandomState(5)
    for _ in range(50):  # same draws in the same order as the reference's np.random calls
        el, az, frontal = sample_training_pose(True, True, 0.3, (-30, 45), (0, 360), rng)
        if ref_rng.random_sample() < 0.3:
            assert (el, az, frontal) == (0.0, 0.0, True)
        else:
            e = ref_rng.uniform(np.radians(-30), np.radians(45)); a = ref_rng.uniform(np.radians(0), np.radians(360))
            assert (el, az, frontal) == (float(e), float(a), False)


def test_ply_known_answer_and_round_trip(tmp_path):
    """Format of GaussianCloud::save_ply / load_ply (renderer.cpp:649-793), byte for byte: the header text, 14
    little-endian floats per vertex, log-scale, (colour - 0.5) / C0, inverse-sigmoid opacity -- and back."""
    from fresnel_amd import io as fio
    g = dict(positions=torch.tensor([[0.5, -1.0, 2.0], [0.0, 0.25, -3.0]]),
             scales=torch.tensor([[0.1, 0.2, 1.0], [1e-9, 2.0, 0.05]]),
             rotations=torch.tensor([[1.0, 0.0, 0.0, 0.0], [0.5, 0.5, -0.5, 0.5]]),
             colors=torch.tensor([[0.5, 1.0, 0.0], [0.25, 0.75, 0.6]]),
             opacities=torch.tensor([0.8, 0.1]))
    path = str(tmp_path / "cloud.ply")
    fio.save_ply(path, g)
    raw = open(path, "rb").read()
    header = ("ply\nformat binary_little_endian 1.0\nelement vertex 2\nproperty float x\nproperty float y\n"
              "property float z\nproperty float scale_0\nproperty float scale_1\nproperty float scale_2\n"
              "property float rot_0\nproperty float rot_1\nproperty float rot_2\nproperty float rot_3\n"
              "property float f_dc_0\nproperty float f_dc_1\nproperty float f_dc_2\nproperty float opacity\nend_header\n")
    assert raw.startswith(header.encode()) and len(raw) == len(header) + 2 * 14 * 4
    v = np.frombuffer(raw[len(header):], dtype="<f4").reshape(2, 14)
    C0 = 0.28209479177387814
    assert np.allclose(v[0], [0.5, -1.0, 2.0, np.log(0.1), np.log(0.2), 0.0, 1, 0, 0, 0, 0.0, 0.5 / C0, -0.5 / C0,
                              np.log(0.8 / 0.2)], rtol=1e-6, atol=1e-6)
    assert np.isclose(v[1, 3], np.log(1e-7), rtol=1e-6)  # scale clamped at 1e-7 before the log
    back = fio.load_ply(path)
    assert torch.allclose(back["positions"], g["positions"]) and torch.allclose(back["rotations"], g["rotations"])
    assert torch.allclose(back["scales"][0], g["scales"][0], rtol=1e-6) and torch.allclose(back["colors"], g["colors"], atol=1e-6)
    assert torch.allclose(back["opacities"], g["opacities"], atol=1e-6)
    assert fio.load_gaussians(path)["positions"].shape == (2, 3)
    # CRLF header, colours outside [0,1] are clamped on load, saturated opacity stays finite
    body = np.zeros((1, 14), "<f4"); body[0, 10:13] = [10.0, -10.0, 0.0]; body[0, 13] = 100.0
    p2 = str(tmp_path / "crlf.ply")
    open(p2, "wb").write(header.replace("vertex 2", "vertex 1").replace("\n", "\r\n").encode() + body.tobytes())
    b2 = fio.load_ply(p2)
    assert b2["colors"].tolist() == [[1.0, 0.0, 0.5]] and float(b2["opacities"][0]) == 1.0
    # errors: truncated body, missing header end, no vertices (the C++ returns false)
    open(str(tmp_path / "short.ply"), "wb").write(raw[:-8])
    with pytest.raises(ValueError):
        fio.load_ply(str(tmp_path / "short.ply"))
    open(str(tmp_path / "nohdr.ply"), "wb").write(b"ply\nelement vertex 3\n")
    with pytest.raises(ValueError):
        fio.load_ply(str(tmp_path / "nohdr.ply"))
    with pytest.raises(ValueError):
        fio.load_gaussians(str(tmp_path / "cloud.obj"))
    # the 14-float binary written by the Python side is what the PLY path reads back after conversion
    pb = str(tmp_path / "cloud.bin")
    fio.save_gaussians_to_binary(pb, g)
    assert torch.equal(fio.load_gaussians(pb)["scales"], g["scales"])


def test_train_harness_reads_data_dir(g12, tmp_path):
    """--data_dir is honoured: make_dataset returns the ImageDataset when the directory holds images, the synthetic
    stand-in otherwise (the flag used to be parsed and ignored)."""
    from fresnel_amd.train import SyntheticDataset, TrainingConfig, make_dataset
    from fresnel_amd.data import ImageDataset
    for rel in g12["ds_files"]:
        full = tmp_path / "d" / str(rel)
        full.parent.mkdir(parents=True, exist_ok=True)
        full.write_bytes(g12["file:" + str(rel)].tobytes())
    cfg = TrainingConfig(data_dir=str(tmp_path / "d"), image_size=24, feature_dim=4, batch_size=2)
    ds, n = make_dataset(cfg, log=lambda *a: None)
    assert isinstance(ds, ImageDataset) and n == 3
    images, feats, depth = ds.batch([0, 1], "cpu")
    assert images.shape == (2, 3, 24, 24) and feats.shape == (2, 37, 37, 4)
    empty = tmp_path / "empty"
    empty.mkdir()
    ds2, n2 = make_dataset(TrainingConfig(data_dir=str(empty), batch_size=2, steps_per_epoch=3), log=lambda *a: None)
    assert isinstance(ds2, SyntheticDataset) and n2 == 6

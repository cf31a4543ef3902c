"""CPU checks: the restatement (oracle/) against the committed golden vectors, and its internal
properties.  The reference holds no tests, fixtures or expected outputs for this path and cannot
be built here, so the vectors were produced by the restatement itself (tests/golden/make_golden.py):
parity with the reference is unpinned, these tests pin the restatement against change."""
import math
import os
import sys

import numpy as np
import pytest

import conftest as cf


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(cf.GOLDEN, "golden.npz"))


def test_lut_cases_match_golden(oracle, golden):
    for name, updates in cf.lut_cases().items():
        lut = np.zeros((256, 4), np.float32)
        for graphs in updates:
            assert oracle.lut_update(lut, graphs) == 0
        assert cf.same_bits(lut, golden["lut_" + name]), name


def test_lut_quirks(oracle):
    # main.cpp:974-992: the two default end points sit at intensity 0 -> all-zero table
    lut = np.zeros((256, 4), np.float32)
    oracle.lut_update(lut, cf.lut_cases()["default_two_points"][0])
    assert not lut.any()
    # truncating loop bound (main.cpp:665): the last table entry is never written
    lut = np.zeros((256, 4), np.float32)
    oracle.lut_update(lut, cf.TF_GRAPHS)
    assert lut[254, 3] > 0.59 and lut[255, 3] == 0.0
    assert np.all(lut[:64, 3] == 0.0) and lut[64, 3] > 0.0
    # entries not covered keep their old value (main.cpp:110, 669-680)
    lut[:] = 7.0
    oracle.lut_update(lut, cf.TF_GRAPHS)
    assert np.all(lut[255] == 7.0)
    # duplicate x: total - 1 == 0 -> 0/0 (main.cpp:667)
    lut = np.zeros((256, 4), np.float32)
    oracle.lut_update(lut, cf.lut_cases()["duplicate_x_nan"][0])
    assert np.isnan(lut[:, 0]).sum() == 1 and np.isnan(lut[255, 0])


def test_control_point_intensity(oracle):
    assert oracle.control_point_intensity(0.0, 300.0) == 1.0
    assert oracle.control_point_intensity(299.0, 300.0) == 0.0


def test_camera_matches_golden(oracle, golden):
    for cname, (h, v) in cf.CAMERAS.items():
        for (w, hgt) in ((512, 512), (1280, 720)):
            vp, m = oracle.camera(h, v, w, hgt)
            assert cf.same_bits(vp, golden["cam_%s_%dx%d_vp" % (cname, w, hgt)])
            assert cf.same_bits(m, golden["cam_%s_%dx%d_m" % (cname, w, hgt)])


def test_camera_default_pose(oracle):
    # default pose (main.cpp:68-77,182-202): side (1,0,0) up (0,0,1) forward (0,1,0) location (0,-5,-.5), height 4.5
    vp, m = oracle.camera(0.0, 0.0, 1280, 720)
    vp = vp.reshape(5, 4)
    assert vp[0, 0] == 4.5
    np.testing.assert_allclose(vp[1], [0, -5, -0.5, 1], atol=1e-6)
    np.testing.assert_allclose(vp[2, :3], [1, 0, 0], atol=1e-6)
    np.testing.assert_allclose(vp[3, :3], [0, 0, 1], atol=1e-6)
    np.testing.assert_allclose(vp[4, :3], [0, 1, 0], atol=1e-6)
    m = m.reshape(4, 4).T  # rows
    corner = m @ np.array([0, 0, 0, 1], np.float32)          # pixel (0,0): top-left of the 1x1 image plane
    np.testing.assert_allclose(corner[:3], [-0.5, -0.5, 0.0], atol=1e-6)
    far = m @ np.array([1279, 719, 0, 1], np.float32)
    np.testing.assert_allclose(far[:3], [0.5, -0.5, -1.0], atol=1e-5)
    # rotating keeps the camera on a sphere around the volume centre (0,0,-.5)
    for h, v in cf.CAMERAS.values():
        loc = oracle.camera(h, v, 64, 64)[0][4:7]
        assert abs(np.linalg.norm(loc - np.array([0, 0, -0.5])) - 5.0) < 1e-5


def test_frames_match_golden(oracle, golden):
    lut = golden["lut_four_points"]
    for dtype, n, tag in ((np.uint8, 32, "u8_32"), (np.uint16, 64, "u16_64")):
        vol, mx = oracle.normalize(oracle.phantom(n, dtype))
        for cname, (h, v) in cf.CAMERAS.items():
            vp, m = oracle.camera(h, v, 64, 64)
            img = oracle.render(vol, lut, vp[4:8], m, 64, 64, 200)
            assert np.array_equal(img, golden["frame_%s_%s" % (tag, cname)]), (tag, cname)
            assert np.all(img[..., 3] == 255)


def test_shaded_frames_match_golden(oracle, golden):
    lut = golden["lut_four_points"]
    vol, _ = oracle.normalize(oracle.phantom(32, np.uint8))
    for cname, (h, v) in cf.CAMERAS.items():
        vp, m = oracle.camera(h, v, 64, 64)
        for sname, scene in cf.SCENES.items():
            img = oracle.render(vol, lut, vp[4:8], m, 64, 64, 200, scene=scene)
            assert np.array_equal(img, golden["shaded_u8_32_%s_%s" % (cname, sname)]), (cname, sname)
            assert not np.array_equal(img, golden["frame_u8_32_" + cname]), (cname, sname)


def test_shading_terms(oracle, golden):
    lut = golden["lut_four_points"]
    vol, _ = oracle.normalize(oracle.phantom(32, np.uint8))
    vp, m = oracle.camera(*cf.CAMERAS["h30_v20"], 48, 40)
    plain = oracle.render(vol, lut, vp[4:8], m, 48, 40, 60)
    # all three weights zero: the lines add 0 to every colour (VolumeRendering.comp:140-151)
    assert np.array_equal(oracle.render(vol, lut, vp[4:8], m, 48, 40, 60, scene=(0.0, 0.0, 0.0, 50.0)), plain)
    # ambient only, everything opaque: the front-most tap's colour + 0.3 * ambient reaches the pixel
    # (scene.ambient * ambientLight, :134,140); a constant volume has gradient 0 away from its faces
    const = np.full((8, 8, 8), 0.5, np.float32)
    flat = np.zeros((256, 4), np.float32)
    flat[:, 0] = 0.25
    flat[:, 3] = 1.0
    vp, m = oracle.camera(0.0, 0.0, 33, 33)
    img = oracle.render(const, flat, vp[4:8], m, 33, 33, 10, border=(0.25, 0.0, 0.0, 1.0), scene=(1.0, 0.0, 0.0, 1.0))
    assert tuple(img[16, 16]) == (140, 76, 76, 255)      # round(0.55 * 255), round(0.3 * 255)
    # a flat neighbourhood (gradient 0) gets the ambient term only, never NaN: diffuse and specular on
    inner = np.full((16, 16, 16), 0.5, np.float32)
    img = oracle.render(inner, flat, vp[4:8], m, 33, 33, 0, border=(0.25, 0.0, 0.0, 1.0), scene=(1.0, 1.0, 1.0, 100.0))
    assert img[16, 16, 1] >= 76


def test_render_rows_and_threads_agree(oracle, golden):
    lut = golden["lut_four_points"]
    vol, _ = oracle.normalize(oracle.phantom(32, np.uint8))
    vp, m = oracle.camera(*cf.CAMERAS["h30_v20"], 48, 40)
    full = oracle.render(vol, lut, vp[4:8], m, 48, 40, 50, threads=1)
    part = oracle.render(vol, lut, vp[4:8], m, 48, 40, 50, rows=(10, 30))
    assert np.array_equal(part[10:30], full[10:30]) and not part[:10].any()


def test_empty_transfer_function_gives_black(oracle):
    vol, _ = oracle.normalize(oracle.phantom(16, np.uint8))
    vp, m = oracle.camera(0.0, 0.0, 32, 32)
    img = oracle.render(vol, np.zeros((256, 4), np.float32), vp[4:8], m, 32, 32, 20)
    assert not img[..., :3].any() and np.all(img[..., 3] == 255)


def test_back_tap_is_not_alpha_weighted(oracle):
    # VolumeRendering.comp:189-191: a constant volume with alpha 0 everywhere shows the back tap's colour
    vol = np.full((8, 8, 8), 0.5, np.float32)
    lut = np.zeros((256, 4), np.float32)
    lut[:, 0] = 0.25
    vp, m = oracle.camera(0.0, 0.0, 33, 33)
    img = oracle.render(vol, lut, vp[4:8], m, 33, 33, 10)
    assert img[16, 16, 0] == 64 and img[16, 16, 1] == 0   # round(0.25 * 255) = 64 (63.75)


def test_phantom_properties(oracle):
    for dtype, dmax in ((np.uint8, 255), (np.uint16, 65535)):
        p = oracle.phantom(48, dtype)
        assert p.dtype == dtype and p.max() <= dmax and p.max() > dmax * 0.99
        assert 0.6 < float((p == 0).mean()) < 0.8          # exterior is exactly zero
        assert np.array_equal(p, oracle.phantom(48, dtype))
        assert not np.array_equal(p, oracle.phantom(48, dtype, seed=1))


def test_normalize_matches_numpy(oracle):
    rng = np.random.default_rng(1)
    v = rng.integers(0, 3273, size=(5, 7, 9)).astype(np.uint16)
    out, mx = oracle.normalize(v)
    assert mx == float(v.max())
    assert cf.same_bits(out, v.astype(np.float32) / np.float32(mx))


def test_histogram(oracle):
    rng = np.random.default_rng(2)
    v = rng.random(5000).astype(np.float32)
    h = oracle.histogram(v)
    assert h.shape == (100,) and h.max() == 1.0 and h.min() > 0.0
    s = np.sort(v)
    step = np.float32(1.0) / np.float32(100)
    for i in (0, 17, 99):
        lo = np.searchsorted(s, np.float32(step * np.float32(i)), side="left")
        hi = np.searchsorted(s.astype(np.float64), float(step) * (i + 0.9), side="right")
        assert h[i] * max(1, round(float(h.max()))) >= 0
        assert hi >= lo


def test_slice_loader_roundtrip(oracle, tmp_path):
    rng = np.random.default_rng(3)
    vol = rng.integers(0, 4000, size=(5, 16, 16)).astype(np.uint16)
    order = [3, 1, 5, 2, 4]                      # written in scrambled order, sorted by extension number
    for k in order:
        vol[k - 1].astype(">u2").tofile(tmp_path / ("slice.%d" % k))
    got, mx = oracle.load_slices(str(tmp_path))
    assert got.shape == (5, 16, 16) and mx == float(vol.max())
    assert cf.same_bits(got, vol.astype(np.float32) / np.float32(mx))
    # files without extension are ordered by their whole name (main.cpp:128)
    d2 = tmp_path / "noext"
    d2.mkdir()
    for k in (10, 9, 11):
        np.full((4, 4), k, ">u2").tofile(d2 / str(k))
    got2, _ = oracle.load_slices(str(d2))
    assert [int(round(s[0, 0] * 11)) for s in got2] == [9, 10, 11]


@pytest.mark.skipif(not os.path.isdir("/root/reference/Applications/VolumeRendering/Dependencies/VolumeData"),
                    reason="reference datasets are only present in the build container")
def test_reference_dataset_loads(oracle):
    vol, mx = oracle.load_slices("/root/reference/Applications/VolumeRendering/Dependencies/VolumeData")
    assert vol.shape == (113, 256, 256) and mx == 3272.0   # CThead: 113 slices of 256x256, maximum 3272


@pytest.mark.skipif(not os.path.isdir("/root/reference/Applications/VolumeRendering/Dependencies/VolumeData"),
                    reason="reference datasets are only present in the build container")
def test_reference_dataset_frame_matches_golden(oracle):
    # the only real-data case: the dataset stays where the reference is, the committed fixture is the frame
    sys.path.insert(0, cf.GOLDEN)
    import make_cthead
    vol, _ = oracle.load_slices(make_cthead.DATA)
    want = np.load(os.path.join(cf.GOLDEN, "cthead_frame.npz"))["frame"]
    assert np.array_equal(make_cthead.render(vol), want)


def test_cthead_block_frames_match_golden(oracle, tmp_path):
    """The committed block of the reference's CThead slices through the checker's slice loader, at the
    reference's window size, sample count and default camera: the frames in cthead_crop.npz."""
    sys.path.insert(0, cf.GOLDEN)
    import make_cthead_crop as mk
    fx = np.load(os.path.join(cf.GOLDEN, "cthead_crop.npz"))
    assert fx["slices"].dtype == np.dtype(">u2") and fx["slices"].shape == (113, 88, 88)
    mk.write_slices(fx["slices"], str(tmp_path))
    frames, shape, mx = mk.frames_of(str(tmp_path))
    assert shape == (113, 88, 88) and mx == float(fx["max_value"]) == 3272.0
    for name, frame in frames.items():
        assert np.array_equal(frame, fx[name]), name
    assert not fx["frame_default_graphs"][..., :3].any() and fx["frame_transparent"][..., :3].any()
    assert not np.array_equal(fx["frame_transparent"], fx["frame_opaque"])

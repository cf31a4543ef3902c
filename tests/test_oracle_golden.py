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


def test_lut_against_a_numpy_transliteration(oracle):
    """updateTransferFunction (main.cpp:643-686) written out once more, statement by statement, in numpy float32
    scalars (every operation rounded to float as the C++ expression rounds it: x / back.x, times 255.0f, the
    float pixel count, k / (count - 1.0f), the float pixel index truncated for the subscript).  The checker's
    table has to equal it bit for bit on every case, the ones that keep entries from earlier calls included."""
    f = np.float32

    def update(table, graphs):
        for channel, graph in enumerate(graphs):
            if not graph:
                continue
            back = f(graph[-1][0])
            points = [(f(f(f(x) / back) * f(255.0)), f(v)) for x, v in graph]          # :649-658
            for j in range(1, len(points)):                                               # :660-662
                direction = f(points[j][1] - points[j - 1][1])                            # :663
                total = f(f(points[j][0] - points[j - 1][0]) + f(1.0))                    # :664
                for k in range(int(total) if total >= 1 else 0):                          # :665, exclusive
                    value = f(points[j - 1][1] + f(direction * f(f(k) / f(total - f(1.0)))))   # :667
                    pixel = f(points[j - 1][0] + f(k))                                    # :668
                    table[int(pixel), channel] = value                                    # :669-680

    with np.errstate(invalid="ignore", divide="ignore"):
        for name, updates in cf.lut_cases().items():
            mine = np.zeros((256, 4), np.float32)
            theirs = np.zeros((256, 4), np.float32)
            for graphs in updates:
                update(mine, graphs)
                assert oracle.lut_update(theirs, graphs) == 0
            same = (mine.view(np.uint32) == theirs.view(np.uint32)) | (np.isnan(mine) & np.isnan(theirs))
            assert same.all(), (name, np.argwhere(~same)[:4])


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


def test_camera_against_an_independent_rotation_library(oracle):
    """The checker restates glm (rotate, translate, column-major products) by hand; here the same uniforms are
    formed in float64 from the reference's statements (main.cpp:68-77, 182-202, 737-799) with scipy's axis-angle
    rotation standing in for glm::rotate -- the right-handed rotation by `angle` about `axis`, which is what both
    document -- and numpy's matrix product.  Agreement to float32 rounding says the hand-written restatement has
    glm's conventions (sign of the angle, order of the products, columns versus rows) right."""
    from scipy.spatial.transform import Rotation

    def translate(v):
        m = np.eye(4)
        m[:3, 3] = v
        return m

    def rotate(angle, axis):
        m = np.eye(4)
        m[:3, :3] = Rotation.from_rotvec(angle * np.asarray(axis, np.float64)).as_matrix()
        return m

    start = np.eye(4)
    start[:3, 2] = [0, 0, -1]                                   # side (1,0,0), up (0,1,0), forward (0,0,-1)  (:70-76)
    pose = translate([0, -5.0, -0.5]) @ rotate(np.radians(90.0), [1, 0, 0]) @ start      # :193-196
    centre = np.array([0, 0, -0.5])                                                       # :68
    for hr, vr_ in cf.CAMERAS.values():
        for (w, hgt) in ((512, 512), (1280, 720), (3840, 2160)):
            turn = rotate(hr, [0, 0, 1]) @ rotate(vr_, [1, 0, 0])                         # :752-755
            basis = pose.copy()
            basis[:, 3] = [0, 0, 0, 1]
            basis = turn @ basis
            location = centre + (turn @ np.append(pose[:3, 3] - centre, 1.0))[:3]         # :756-757
            frame = basis.copy()
            frame[:, 3] = [0, 0, 0, 1]
            to_world = (translate(location + 4.5 * basis[:3, 2]) @ frame @ translate([-0.5, 0.5, 0]) @
                        np.diag([1.0, -1.0, 1.0, 1.0]) @ np.diag([1.0 / (w - 1.0), 1.0 / (hgt - 1.0), 1.0, 1.0]))   # :771-794
            vp, m = oracle.camera(hr, vr_, w, hgt)
            np.testing.assert_allclose(m.reshape(4, 4).T, to_world, atol=2e-6, rtol=0)
            vp = vp.reshape(5, 4)
            np.testing.assert_allclose(vp[0], [4.5, 0, 0, 0], atol=0)
            np.testing.assert_allclose(vp[1], np.append(location, 1.0), atol=2e-6)
            for k in range(3):
                np.testing.assert_allclose(vp[2 + k], basis[:, k], atol=2e-6)
            # a pixel's ray: the image plane is one unit wide and high, 4.5 from the apex, pixel (0, 0) top left
            centre_pixel = to_world @ np.array([(w - 1) / 2.0, (hgt - 1) / 2.0, 0.0, 1.0])
            np.testing.assert_allclose(centre_pixel[:3], location + 4.5 * basis[:3, 2], atol=1e-9)


def test_frames_match_golden(oracle, golden):
    lut = golden["lut_four_points"]
    for dtype, n, tag in ((np.uint8, 32, "u8_32"), (np.uint16, 64, "u16_64")):
        vol, mx = oracle.normalize(oracle.phantom(n, dtype))
        for cname, (h, v) in cf.CAMERAS.items():
            vp, m = oracle.camera(h, v, 64, 64)
            img = oracle.render(vol, lut, vp[4:8], m, 64, 64, 200)
            assert np.array_equal(img, golden["frame_%s_%s" % (tag, cname)]), (tag, cname)
            assert np.all(img[..., 3] == 255)


def test_frames_against_a_library_sampler(oracle, golden):
    """The shader's main() (VolumeRendering.comp:154-206, slab test :50-96) written a second time, vectorised over the
    pixels in float64, with torch's grid_sample standing in for both texture() calls: linear filtering between texel
    centres at (i + 0.5) / size with a zero border is what `align_corners=False, padding_mode="zeros"` computes and what
    the sampler of main.cpp:365-382 asks Vulkan for.  The checker's scalar float32 loop has to agree with it up to
    rounding: at most one unit in a byte, and nearly all bytes equal.  This does not pin the checker to the
    reference (nothing can here); it pins its texel addressing, border, blend order and back tap to an
    implementation that shares no code with it."""
    import torch
    import torch.nn.functional as F

    lut = np.asarray(golden["lut_four_points"], np.float64).reshape(256, 4)
    table = torch.from_numpy(lut.T.copy()).reshape(1, 4, 1, 256)
    w = hgt = 64
    samples = 200
    for dtype, n in ((np.uint8, 32), (np.uint16, 64)):
        vol, _ = oracle.normalize(oracle.phantom(n, dtype))
        texture = torch.from_numpy(vol.astype(np.float64)).reshape(1, 1, n, n, n)
        for cname, (hr, vr_) in cf.CAMERAS.items():
            vp, m = oracle.camera(hr, vr_, w, hgt)
            to_world = m.astype(np.float64).reshape(4, 4).T
            eye = vp[4:7].astype(np.float64)
            px, py = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(hgt, dtype=np.float64))
            pixel = np.stack([px, py, np.zeros_like(px), np.ones_like(px)], -1) @ to_world.T
            d = pixel[..., :3] - eye
            d /= np.linalg.norm(d, axis=-1, keepdims=True)
            lo, hi = np.array([-0.5, -0.5, -1.0]), np.array([0.5, 0.5, 0.0])
            with np.errstate(divide="ignore", invalid="ignore"):
                t1, t2 = (lo - eye) / d, (hi - eye) / d
            tmin = np.minimum(t1, t2).max(-1)
            tmax = np.maximum(t1, t2).min(-1)
            hit = ~((tmin > tmax) | (tmax < 0))
            step = (tmax - tmin) / (samples + 1)
            colour = np.zeros((hgt, w, 3))
            for t in range(samples + 2):
                at = eye + (tmax - t * step)[..., None] * d
                uvw = np.abs(at - np.array([-0.5, -0.5, 0.0]))
                grid = torch.from_numpy(np.nan_to_num(2.0 * uvw - 1.0)).reshape(1, 1, hgt, w, 3)
                value = F.grid_sample(texture, grid, mode="bilinear", padding_mode="zeros", align_corners=False).reshape(hgt, w)
                at_table = torch.stack([2.0 * value - 1.0, torch.zeros_like(value)], -1).reshape(1, hgt, w, 2)
                rgba = F.grid_sample(table, at_table, mode="bilinear", padding_mode="zeros", align_corners=False)[0].permute(1, 2, 0).numpy()
                if t == 0:
                    colour = rgba[..., :3].copy()
                else:
                    a = rgba[..., 3:4]
                    colour = a * rgba[..., :3] + (1.0 - a) * colour
            colour[~hit] = 0.0
            want = np.rint(np.clip(colour, 0.0, 1.0) * 255.0).astype(np.int32)
            got = oracle.render(vol, lut.astype(np.float32), vp[4:8], m, w, hgt, samples)[..., :3].astype(np.int32)
            diff = np.abs(got - want)
            print("%s %d^3: %.4f of the bytes equal, largest difference %d" % (cname, n, (diff == 0).mean(), diff.max()))
            assert diff.max() <= 1, (cname, n, int(diff.max()))
            assert (diff == 0).mean() >= 0.995, (cname, n, float((diff == 0).mean()))
            assert hit.any() and not hit.all()


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
    """prepareHistogramVolumeData (main.cpp:166-181) once more in numpy: sort, lower_bound of the float step * i,
    upper_bound of the double step * (i + 0.9), the difference as a float, divided by the largest count.  All
    100 bins bit for bit, on continuous values and on normalised integers (many ties, values on bin edges)."""
    rng = np.random.default_rng(2)
    u16 = (rng.random(20000) ** 2 * 4095).astype(np.uint16)
    u8 = rng.integers(0, 256, 5000).astype(np.uint8)
    cases = [rng.random(5000).astype(np.float32), oracle.normalize(u16.reshape(20, 25, 40))[0].ravel(),
             oracle.normalize(u8.reshape(10, 20, 25))[0].ravel(), np.linspace(0, 1, 101, dtype=np.float32)]
    step = np.float32(1.0) / np.float32(100)                                                  # :170
    for v in cases:
        got = oracle.histogram(v)
        s = np.sort(v)                                                                        # :171
        counts = np.zeros(100, np.float32)
        for i in range(100):
            lo = np.searchsorted(s, np.float32(step * np.float32(i)), side="left")            # :174
            hi = np.searchsorted(s.astype(np.float64), float(step) * (i + 0.9), side="right")  # :175, double
            counts[i] = np.float32(hi - lo)                                                   # :176
        want = counts / np.float32(int(counts.max()))                                         # :178-180
        assert got.shape == (100,) and cf.same_bits(got, want)
        assert got.max() == 1.0


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

"""GPU parity tests: every check calls the HIP path through the C ABI (libvr_hip.so) and compares
with the CPU checker on the same inputs, or with the committed golden vectors.

Bars: transfer-function table, camera uniforms, phantom and normalised voxels bit-exact; frames
at most 1/255 per channel off with at least 99.9 % of the pixels identical (the kernels are built
to be bit-identical to the checker, which the tests also report and assert)."""
import math
import os

import numpy as np
import pytest

import conftest as cf

pytestmark = pytest.mark.gpu

MAX_DELTA = 1          # 1/255 per RGBA8 channel
MIN_EXACT = 0.999      # fraction of identical pixels


@pytest.fixture(scope="module")
def vr():
    import vulkanrenderer_amd as m
    return m


@pytest.fixture(scope="module")
def torch():
    import torch as t
    assert t.cuda.is_available()
    return t


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(cf.GOLDEN, "golden.npz"))


@pytest.fixture()
def ctx(vr):
    c = vr.VolumeRenderer(0)
    yield c
    c.close()


def check_frame(got, want, what, exact_required=True):
    worst, exact = cf.frame_report(got, want)
    print("%s: max|delta|=%d exact=%.5f" % (what, worst, exact))
    assert worst <= MAX_DELTA and exact >= MIN_EXACT, what
    if exact_required:
        assert exact == 1.0, what
    assert np.all(got[..., 3] == 255)


# ---- transfer function (bit-exact) ------------------------------------------------------------

def test_lut_bit_exact_all_cases(ctx, oracle, golden):
    for name, updates in cf.lut_cases().items():
        want = np.zeros((256, 4), np.float32)
        for k, graphs in enumerate(updates):
            oracle.lut_update(want, graphs)
            ctx.update_transfer_function(graphs, reset_state=(k == 0))
            assert cf.same_bits(ctx.transfer_function(), want), (name, k)
        assert cf.same_bits(ctx.transfer_function(), golden["lut_" + name]), name


def test_lut_random_graphs_bit_exact(ctx, oracle):
    rng = np.random.default_rng(11)
    want = np.zeros((256, 4), np.float32)
    ctx.update_transfer_function([[], [], [], []], reset_state=True)
    for trial in range(40):
        graphs = []
        for c in range(4):
            n = int(rng.integers(2, 17))
            xs = np.sort(rng.integers(0, 483, size=n).astype(np.float32))
            if rng.random() < 0.5:
                xs += rng.random(n).astype(np.float32)     # fractional pixel positions
                xs = np.sort(xs)
            xs[0] = 0.0
            if xs[-1] <= 0:
                xs[-1] = 482.0
            ys = rng.random(n).astype(np.float32)
            graphs.append(list(zip(xs.tolist(), ys.tolist())))
        oracle.lut_update(want, graphs)
        ctx.update_transfer_function(graphs)
        assert cf.same_bits(ctx.transfer_function(), want), trial


def test_lut_rejects_bad_graphs(ctx, vr):
    with pytest.raises(vr.VrError):
        ctx.update_transfer_function([[(10.0, 0.1), (5.0, 0.2)], [], [], []])      # not ascending
    with pytest.raises(vr.VrError):
        ctx.update_transfer_function([[(0.0, 0.1), (0.0, 0.2)], [], [], []])       # x_last == 0
    with pytest.raises(vr.VrError):
        ctx.update_transfer_function([[(-1.0, 0.1), (5.0, 0.2)], [], [], []])


# ---- camera -----------------------------------------------------------------------------------

def test_camera_uniforms_bit_exact(ctx, oracle, golden):
    for cname, (h, v) in cf.CAMERAS.items():
        ctx.set_camera(h, v)
        for (w, hgt) in ((512, 512), (1280, 720)):
            vp, m = ctx.pixel_transformation_uniforms(w, hgt)
            assert cf.same_bits(vp, golden["cam_%s_%dx%d_vp" % (cname, w, hgt)])
            assert cf.same_bits(m, golden["cam_%s_%dx%d_m" % (cname, w, hgt)])


# ---- ingest ------------------------------------------------------------------------------------

@pytest.mark.parametrize("dtype,n", [(np.uint8, 40), (np.uint16, 37)])
def test_phantom_device_equals_checker(ctx, oracle, torch, dtype, n):
    tdt = torch.uint8 if dtype == np.uint8 else torch.int16
    buf = torch.empty(n ** 3, dtype=tdt, device="cuda")
    torch.cuda.synchronize()
    ctx.phantom_generate_device(buf.data_ptr(), np.dtype(dtype).itemsize, n)
    ctx.synchronize()
    got = buf.cpu().numpy().view(dtype).reshape(n, n, n)
    assert np.array_equal(got, oracle.phantom(n, dtype))


@pytest.mark.parametrize("dtype,shape", [(np.uint8, (13, 21, 30)), (np.uint16, (9, 16, 23)), (np.uint16, (1, 1, 1))])
def test_normalised_voxels_bit_exact(ctx, oracle, dtype, shape):
    rng = np.random.default_rng(5)
    hi = 256 if dtype == np.uint8 else 3273
    vox = rng.integers(0, hi, size=shape).astype(dtype)
    vox.flat[0] = hi - 1
    ctx.upload_volume(vox)
    want, mx = oracle.normalize(vox)
    assert ctx.volume_info()["max_value"] == mx
    assert cf.same_bits(ctx.read_volume_normalized(), want)


@pytest.mark.parametrize("layers", [1, 3, 4])
def test_ingest_strips_and_layers(ctx, oracle, monkeypatch, tmp_path, layers):
    """The re-tiling kernel on shapes that take its aligned 16-byte loads (row length a multiple of 8) and its
    one-by-one loads, several strips of 16 bricks with a short last one, and workgroups that walk several brick
    layers with the shared voxel slice kept in registers (small volumes do that only when told to)."""
    monkeypatch.setenv("VR_TEST_INGEST_LAYERS", str(layers))
    rng = np.random.default_rng(11 + layers)
    for dtype, shape in ((np.uint16, (37, 26, 272)), (np.uint8, (29, 19, 144)), (np.uint16, (21, 9, 141)), (np.uint8, (70, 3, 8))):
        hi = 256 if dtype == np.uint8 else 60000
        vox = rng.integers(0, hi, size=shape).astype(dtype)
        ctx.upload_volume(vox)
        want, mx = oracle.normalize(vox)
        assert ctx.volume_info()["max_value"] == mx
        assert cf.same_bits(ctx.read_volume_normalized(), want), (dtype, shape)
    # big-endian slices through the same kernel (main.cpp:137-143)
    vol = rng.integers(0, 4000, size=(19, 40, 40)).astype(np.uint16)
    for k in range(19):
        vol[k].astype(">u2").tofile(tmp_path / ("s.%d" % (k + 1)))
    ctx.load_volume_data(str(tmp_path))
    want, mx = oracle.load_slices(str(tmp_path))
    assert ctx.volume_info()["max_value"] == mx
    assert cf.same_bits(ctx.read_volume_normalized(), want)


def test_second_volume_of_the_same_shape(ctx, oracle):
    """A new volume with the brick grid and voxel type of the old one moves into the old one's buffers: nothing of
    the old volume may survive -- voxels, maximum, macrocells, occupancy -- and a frame in fast mode is the
    checker's frame of the new volume.  Then another shape, and another voxel type (new buffers)."""
    rng = np.random.default_rng(3)
    first = oracle.phantom(40, np.uint16)
    second = (rng.integers(0, 30000, size=first.shape) * (rng.random(first.shape) < 0.3)).astype(np.uint16)
    second[5:9, 5:9, 5:9] = 41234
    ctx.update_transfer_function(cf.TF_GRAPHS, reset_state=True)
    ctx.set_camera(*cf.CAMERAS["h30_v20"])
    ctx.set_sampling(100)
    lut = np.zeros((256, 4), np.float32)
    oracle.lut_update(lut, cf.TF_GRAPHS)
    vp, m = oracle.camera(*cf.CAMERAS["h30_v20"], 96, 80)
    for vox in (first, second, first[:, :24], second.astype(np.uint8)):
        ctx.upload_volume(vox)
        want, mx = oracle.normalize(vox)
        assert ctx.volume_info()["max_value"] == mx
        assert cf.same_bits(ctx.read_volume_normalized(), want)
        ref = oracle.render(want, lut, vp[4:8], m, 96, 80, 100)
        for mode in (1, 2, 0):
            ctx.set_mode(mode)
            assert np.array_equal(ctx.render(96, 80), ref), (vox.shape, vox.dtype, mode)


def test_normalisation_exact_for_every_u16_value(ctx, oracle):
    for mx in (65535, 63536, 5119, 3272, 257, 3):
        vox = (np.arange(65536, dtype=np.uint32) % (mx + 1)).astype(np.uint16).reshape(16, 64, 64)
        ctx.upload_volume(vox)
        want, got_mx = oracle.normalize(vox)
        assert got_mx == float(mx)
        assert cf.same_bits(ctx.read_volume_normalized(), want), mx


def test_explicit_max_value(ctx, oracle, vr):
    vox = np.arange(4 * 4 * 4, dtype=np.uint16).reshape(4, 4, 4)
    ctx.upload_volume(vox, max_value=1000.0)
    assert cf.same_bits(ctx.read_volume_normalized(), oracle.normalize(vox, 1000.0)[0])
    with pytest.raises(vr.VrError):
        ctx.upload_volume(vox, max_value=10.0)        # below the data maximum
    with pytest.raises(vr.VrError):
        ctx.upload_volume(np.zeros((4, 4, 4), np.uint8))   # the reference would divide by zero


def test_slice_loader_matches_checker(ctx, oracle, tmp_path):
    rng = np.random.default_rng(7)
    vol = rng.integers(0, 4000, size=(6, 24, 24)).astype(np.uint16)
    for k in (4, 2, 6, 1, 5, 3):
        vol[k - 1].astype(">u2").tofile(tmp_path / ("CThead.%d" % k))
    ctx.load_volume_data(str(tmp_path))
    want, mx = oracle.load_slices(str(tmp_path))
    info = ctx.volume_info()
    assert (info["nz"], info["ny"], info["nx"]) == want.shape and info["max_value"] == mx
    assert cf.same_bits(ctx.read_volume_normalized(), want)


def test_slice_loader_errors(ctx, vr, tmp_path):
    with pytest.raises(vr.VrError):
        ctx.load_volume_data(str(tmp_path / "missing"))
    with pytest.raises(vr.VrError):
        ctx.load_volume_data(str(tmp_path))             # empty directory (main.cpp:158)
    # std::stoi on the key of a name (main.cpp:123-128) throws when it holds no number
    good = np.arange(16, dtype=">u2")
    for bad in (".hidden", "slice.", "slice.abc"):
        d = tmp_path / ("case_" + bad.strip(".").replace(".", "_"))
        d.mkdir()
        good.tofile(d / "CThead.1")
        good.tofile(d / bad)
        with pytest.raises(vr.VrError, match="slice number"):
            ctx.load_volume_data(str(d))
    # ... and takes the leading digits when something follows them ("2x" sorts as 2)
    d = tmp_path / "mixed"
    d.mkdir()
    good.tofile(d / "a.1")
    (np.arange(16, 32).astype(">u2")).tofile(d / "a.2x")
    ctx.load_volume_data(str(d))
    info = ctx.volume_info()
    assert (info["nx"], info["ny"], info["nz"]) == (4, 4, 2) and info["max_value"] == 31.0


def test_failed_upload_leaves_no_volume(vr, oracle, monkeypatch):
    """An allocation that fails in the middle of an upload must leave the context without a volume
    (render answers with a state error), not with a half-built one."""
    vox = oracle.phantom(24, np.uint8)
    for nth in range(1, 11):          # bricks, cells, touched, four pyramid levels, micro, skip, occ_list
        monkeypatch.setenv("VR_TEST_FAIL_ALLOC", str(nth))
        with vr.VolumeRenderer(0) as c:
            monkeypatch.delenv("VR_TEST_FAIL_ALLOC")
            c.update_transfer_function(cf.TF_GRAPHS, reset_state=True)
            with pytest.raises(vr.VrError, match="volume upload"):
                c.upload_volume(vox)
            for mode in (0, 1, 2):
                c.set_mode(mode)
                with pytest.raises(vr.VrError, match="no volume"):
                    c.render(32, 32)
            with pytest.raises(vr.VrError):
                c.volume_info()
            # the hook has fired; the next upload goes through and renders
            c.upload_volume(vox)
            c.set_mode(1)
            c.set_sampling(40)
            assert c.render(32, 32)[..., :3].any()


# ---- frames ------------------------------------------------------------------------------------

def render_pair(ctx, oracle, vox, graphs, cam, w, h, samples, border=(0, 0, 0, 0), mode=0, stats=False):
    ctx.upload_volume(vox)
    ctx.update_transfer_function(graphs, reset_state=True)
    ctx.set_camera(*cam)
    ctx.set_sampling(samples, border)
    ctx.set_mode(mode)
    got = ctx.render(w, h, stats=stats)
    vol, _ = oracle.normalize(vox)
    lut = np.zeros((256, 4), np.float32)
    oracle.lut_update(lut, graphs)
    vp, m = oracle.camera(cam[0], cam[1], w, h)
    want = oracle.render(vol, lut, vp[4:8], m, w, h, samples, border=border)
    return got, want


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_golden_frames(ctx, oracle, golden, mode):
    for dtype, n, tag in ((np.uint8, 32, "u8_32"), (np.uint16, 64, "u16_64")):
        ctx.upload_volume(oracle.phantom(n, dtype))
        ctx.update_transfer_function(cf.TF_GRAPHS, reset_state=True)
        ctx.set_sampling(200, (0, 0, 0, 0))
        ctx.set_mode(mode)
        for cname, cam in cf.CAMERAS.items():
            ctx.set_camera(*cam)
            check_frame(ctx.render(64, 64), golden["frame_%s_%s" % (tag, cname)], "golden %s %s mode %d" % (tag, cname, mode))


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("w,h", [(50, 37), (129, 65), (16, 16)])
def test_ragged_image_sizes(ctx, oracle, mode, w, h):
    got, want = render_pair(ctx, oracle, oracle.phantom(24, np.uint16), cf.TF_GRAPHS, cf.CAMERAS["h30_v20"], w, h, 60, mode=mode)
    check_frame(got, want, "ragged %dx%d" % (w, h))


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_non_cubic_volume_and_border_colour(ctx, oracle, mode):
    rng = np.random.default_rng(9)
    vox = rng.integers(0, 3000, size=(11, 33, 20)).astype(np.uint16)      # nz, ny, nx all different
    for border in ((0, 0, 0, 0), (0, 0, 0, 1), (0.2, 0.4, 0.6, 0.3)):
        got, want = render_pair(ctx, oracle, vox, cf.TF_GRAPHS, cf.CAMERAS["hm135_vm60"], 96, 80, 100, border=border, mode=mode)
        check_frame(got, want, "noise volume border %s" % (border,))


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_config1_full_size(ctx, oracle, mode):
    # BASELINE configs[0]: 256^3 uint8 phantom, 512x512, fixed camera (30, 20 degrees), 4-point TF, 200 samples
    got, want = render_pair(ctx, oracle, oracle.phantom(256, np.uint8), cf.TF_GRAPHS, cf.CAMERAS["h30_v20"], 512, 512, 200,
                            mode=mode, stats=True)
    img, st = got
    check_frame(img, want, "config 1")
    vp, m = oracle.camera(*cf.CAMERAS["h30_v20"], 512, 512)
    assert st["rays"] == 512 * 512 and st["rays_hit"] == oracle.count_hits(vp[4:8], m, 512, 512)
    assert st["samples_nominal"] == st["rays_hit"] * 202
    if mode == 0:
        assert st["samples_evaluated"] == st["samples_nominal"]
    assert 0 < st["bricks_touched"] <= 33 ** 3 and st["algorithmic_bytes"] > 512 * 512 * 4


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_rect_equals_crop_of_frame(ctx, oracle, mode):
    ctx.upload_volume(oracle.phantom(32, np.uint8))
    ctx.update_transfer_function(cf.TF_GRAPHS, reset_state=True)
    ctx.set_camera(*cf.CAMERAS["h30_v20"])
    ctx.set_sampling(80)
    ctx.set_mode(mode)
    full = ctx.render(100, 70)
    for rect in ((0, 0, 100, 70), (16, 16, 48, 32), (37, 5, 99, 61), (99, 69, 100, 70)):
        x0, y0, x1, y1 = rect
        assert np.array_equal(ctx.render(100, 70, rect=rect), full[y0:y1, x0:x1]), rect


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_camera_inside_the_box(ctx, oracle, mode):
    # tMin < 0 is used as is (VolumeRendering.comp:93-95): taps behind the apex are still taken
    vox = oracle.phantom(32, np.uint16)
    ctx.upload_volume(vox)
    ctx.update_transfer_function(cf.TF_GRAPHS, reset_state=True)
    ctx.set_sampling(64)
    ctx.set_mode(mode)
    vp, m = oracle.camera(0.3, 0.2, 64, 64)
    loc = np.array([0.05, -0.1, -0.45, 1.0], np.float32)
    m = m.copy().reshape(4, 4)
    m[3, :3] = loc[:3] + np.float32(0.2) * vp[16:19]       # image plane 0.2 in front of the apex
    m[0, :3] *= np.float32(0.3)
    m[1, :3] *= np.float32(0.3)
    ctx.set_camera_raw(loc, m.ravel())
    got = ctx.render(64, 64)
    vol, _ = oracle.normalize(vox)
    lut = np.zeros((256, 4), np.float32)
    oracle.lut_update(lut, cf.TF_GRAPHS)
    want = oracle.render(vol, lut, loc, m.ravel(), 64, 64, 64)
    check_frame(got, want, "camera inside box")
    assert want[..., :3].any()


@pytest.mark.parametrize("case", ["apex_on_a_pixel", "nan_in_one_row", "infinite_translation"])
def test_rays_that_are_not_numbers(ctx, oracle, case):
    """Raw uniforms that make tap positions NaN: a camera whose apex is the world position of one of its own
    pixels (direction 0 / 0), a NaN in the row of the matrix that gives world y (x and z stay finite, so the slab
    test still finds hits), an infinite translation.  The shader's arithmetic turns such pixels black -- every tap
    NaN, NaN stored as 0 -- or misses the box; the index clamp sends NaN to the border texel.  All modes must agree
    with the checker, and the modes that skip and stop early must not take the border brick such taps index for
    empty space that leaves the colour alone."""
    w, h, samples = 64, 48, 80
    vox = oracle.phantom(32, np.uint16)
    ctx.upload_volume(vox)
    ctx.update_transfer_function(cf.TF_GRAPHS, reset_state=True)
    ctx.set_sampling(samples)
    vp, m = oracle.camera(math.radians(30.0), math.radians(20.0), w, h)
    loc, m = vp[4:8].copy(), m.copy()
    if case == "apex_on_a_pixel":
        x0, y0 = np.float32(23), np.float32(17)
        for c in range(3):      # world position of pixel (23, 17) in the kernel's own operation order
            loc[c] = m[c] * x0 + (m[4 + c] * y0 + m[12 + c])
    elif case == "nan_in_one_row":
        m[1] = np.float32("nan")
    else:
        m[12] = np.float32("inf")
    ctx.set_camera_raw(loc, m)
    vol, _ = oracle.normalize(vox)
    lut = np.zeros((256, 4), np.float32)
    oracle.lut_update(lut, cf.TF_GRAPHS)
    want = oracle.render(vol, lut, loc, m, w, h, samples)
    assert np.all(want[..., 3] == 255)
    if case == "apex_on_a_pixel":
        assert np.all(want[17, 23, :3] == 0)
    for mode in (0, 2, 1):
        ctx.set_mode(mode)
        got = ctx.render(w, h)
        assert np.array_equal(got, want), (case, mode, cf.frame_report(got, want))


def test_device_output_pointer(ctx, oracle, torch):
    ctx.upload_volume(oracle.phantom(32, np.uint8))
    ctx.update_transfer_function(cf.TF_GRAPHS, reset_state=True)
    ctx.set_camera(*cf.CAMERAS["front"])
    ctx.set_sampling(50)
    host = ctx.render(64, 48)
    buf = torch.zeros(48 * 64 * 4, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.render_device(64, 48, buf.data_ptr())
    ctx.synchronize()
    assert np.array_equal(buf.cpu().numpy().reshape(48, 64, 4), host)


def test_call_order_and_argument_errors(vr):
    with vr.VolumeRenderer(0) as c:
        with pytest.raises(vr.VrError):
            c.render(64, 64)                                   # no volume yet
        c.upload_volume(np.ones((4, 4, 4), np.uint8))
        with pytest.raises(vr.VrError):
            c.render(1, 64)                                    # 1/(extent-1), main.cpp:772
        with pytest.raises(vr.VrError):
            c.render(64, 64, rect=(0, 0, 65, 64))
        with pytest.raises(vr.VrError):
            c.set_sampling(-1)
        with pytest.raises(vr.VrError):
            c.set_mode(7)
    h = vr.C.c_void_p()
    assert vr.lib().vr_create(9999, vr.C.byref(h)) != 0


# ---- fast mode specifics -----------------------------------------------------------------------

def test_fast_mode_skips_and_terminates(ctx, oracle):
    vox = oracle.phantom(128, np.uint16)
    got, want = render_pair(ctx, oracle, vox, cf.TF_GRAPHS, cf.CAMERAS["h30_v20"], 256, 256, 443, mode=1, stats=True)
    img, st = got
    check_frame(img, want, "fast 128^3")
    print(st)
    assert st["samples_nominal"] == st["rays_hit"] * 445
    assert st["samples_evaluated"] < 0.25 * st["samples_nominal"]      # skipping + termination do most of the work
    assert st["replay_rays"] < 0.2 * st["rays_hit"]
    # the same frame in reference order with skipping fetches more taps but far fewer than all
    ctx.set_mode(2)
    img2, st2 = ctx.render(256, 256, stats=True)
    assert np.array_equal(img2, img)
    assert st["samples_evaluated"] < st2["samples_evaluated"] < st2["samples_nominal"]


def test_fast_mode_falls_back_when_alpha_leaves_unit_interval(ctx, oracle):
    # intensities are UI-clamped to [0,1] in the reference, but the table is plain floats: with
    # alpha > 1 the interval argument of the fast march does not hold and every ray is re-marched
    graphs = [list(g) for g in cf.TF_GRAPHS]
    graphs[3] = [(0.0, 0.0), (120.0, 0.0), (300.0, 0.4), (483.0, 1.7)]
    got, want = render_pair(ctx, oracle, oracle.phantom(32, np.uint8), graphs, cf.CAMERAS["h30_v20"], 64, 64, 100, mode=1, stats=True)
    img, st = got
    check_frame(img, want, "alpha > 1")
    assert st["replay_rays"] == st["rays_hit"]


def test_nan_in_table(ctx, oracle):
    graphs = cf.lut_cases()["duplicate_x_nan"][0]
    for mode in (0, 1, 2):
        got, want = render_pair(ctx, oracle, oracle.phantom(32, np.uint8), graphs, cf.CAMERAS["front"], 48, 48, 60, mode=mode)
        check_frame(got, want, "NaN table mode %d" % mode)


def test_opaque_border_colour_fast(ctx, oracle):
    # hardware that honours "opaque black" gives the border alpha 1: air becomes half opaque
    got, want = render_pair(ctx, oracle, oracle.phantom(48, np.uint16), cf.TF_GRAPHS, cf.CAMERAS["h30_v20"], 80, 80, 120,
                            border=(0, 0, 0, 1), mode=1)
    check_frame(got, want, "opaque border fast")


def test_translucent_table_fast(ctx, oracle):
    # low alphas everywhere: rays rarely terminate early, long accumulation chains
    xs = (0.0, 60.0, 300.0, 483.0)
    graphs = [list(zip(xs, c)) for c in ((0.2, .9, 1, 1), (0.1, .3, .8, 1), (0.3, .1, .5, 1), (0.0, 0.004, 0.01, 0.02))]
    got, want = render_pair(ctx, oracle, oracle.phantom(64, np.uint16), graphs, cf.CAMERAS["hm135_vm60"], 96, 96, 300, mode=1, stats=True)
    img, st = got
    check_frame(img, want, "translucent fast")
    print(st)


def test_random_tables_volumes_and_views_all_modes_agree(ctx, oracle):
    """Seeded random cases the hand-written ones do not reach: tables whose alpha graph goes to zero and back
    several times (transparent windows in the middle and at the top of the value range, so the set of transparent
    entries is neither an interval nor monotone), volumes of smooth random blobs with odd sizes, cameras far,
    near and inside, odd image sizes, few and many samples, random border colours.  The reference-order march
    is the checker's (proved elsewhere); here the skipping and the fast march have to give its bytes, and the
    first few cases are checked against the checker itself."""
    rng = np.random.default_rng(20240611)
    lit = 0
    for case in range(60):
        shape = tuple(int(v) for v in rng.integers(9, 70, 3))
        dtype = np.uint16 if case % 3 else np.uint8
        top = 65535 if dtype == np.uint16 else 255
        # a few Gaussian blobs on a coarse random field, some voxels exactly zero
        z, y, x = np.meshgrid(*(np.linspace(-1, 1, n) for n in shape), indexing="ij")
        field = np.zeros(shape)
        for _ in range(int(rng.integers(1, 5))):
            c = rng.uniform(-0.6, 0.6, 3)
            field += rng.uniform(0.3, 1.0) * np.exp(-((z - c[0]) ** 2 + (y - c[1]) ** 2 + (x - c[2]) ** 2) / rng.uniform(0.02, 0.3))
        field += 0.05 * rng.random(shape)
        field[field < rng.uniform(0.03, 0.2)] = 0.0
        vox = (np.clip(field / max(field.max(), 1e-9), 0, 1) * top).astype(dtype)
        if not vox.any():
            vox[0, 0, 0] = 1
        # control graphs: 3 to 8 points, alpha zero at a random subset of them
        n_pts = int(rng.integers(3, 9))
        xs = np.sort(rng.uniform(0.0, 480.0, n_pts))
        xs[0] = 0.0
        graphs = []
        for ch in range(4):
            vals = rng.random(n_pts)
            if ch == 3:
                vals = vals * rng.choice([0.05, 0.4, 1.0])
                vals[rng.random(n_pts) < 0.5] = 0.0
            graphs.append([(float(a), float(b)) for a, b in zip(xs, vals)])
        w, h = int(rng.integers(17, 120)), int(rng.integers(17, 90))
        samples = int(rng.choice([7, 40, 150, 333]))
        border = (0, 0, 0, 0) if case % 4 else tuple(float(v) for v in rng.random(4) * [1, 1, 1, rng.choice([0.0, 0.3])])
        ctx.upload_volume(vox)
        ctx.update_transfer_function(graphs, reset_state=True)
        ctx.set_sampling(samples, border)
        if case % 5 == 4:      # a camera somewhere inside or just outside the box, looking anywhere
            eye = rng.uniform(-0.7, 0.7, 3) + np.array([0.0, 0.0, -0.5])
            rot = np.linalg.qr(rng.normal(size=(3, 3)))[0].astype(np.float32)
            m = np.eye(4, dtype=np.float32)
            m[:3, :3] = rot * np.array([1.0 / (w - 1), 1.0 / (h - 1), 1.0], np.float32)
            m[:3, 3] = eye + rot @ np.array([-0.5, -0.5, 1.0], np.float32)
            ctx.set_camera_raw(tuple(eye) + (1.0,), m.T.ravel())
        else:
            ctx.set_camera(float(rng.uniform(-np.pi, np.pi)), float(rng.uniform(-np.pi / 2, np.pi / 2)))
        ctx.set_mode(0)
        want = ctx.render(w, h)
        lit += int(want[..., :3].any())
        if case < 6:
            vol, _ = oracle.normalize(vox)
            lut = np.zeros((256, 4), np.float32)
            oracle.lut_update(lut, graphs)
            vp, m4 = ctx.pixel_transformation_uniforms(w, h)
            assert np.array_equal(want, oracle.render(vol, lut, vp[4:8], m4, w, h, samples, border=border)), case
        for mode in (2, 1):
            ctx.set_mode(mode)
            got = ctx.render(w, h)
            assert np.array_equal(got, want), (case, mode, shape, samples, cf.frame_report(got, want))
    assert lit >= 40, lit      # most cases show something


def test_fast_mode_gives_way_under_a_translucent_table(ctx, oracle):
    """Low alpha everywhere: no ray becomes opaque, and several in a hundred of the first fast frame's pixels end in
    the replay (one in a thousand under a table that makes rays opaque).
    From the second frame on fast mode renders through the skipping reference-order march: same bytes, no
    more taps than that mode fetches.  A new table re-arms the fast path."""
    xs = (0.0, 60.0, 300.0, 483.0)
    translucent = [list(zip(xs, c)) for c in ((0.2, .9, 1, 1), (0.1, .3, .8, 1), (0.3, .1, .5, 1), (0.0, 0.004, 0.01, 0.02))]
    vox = oracle.phantom(96, np.uint16)
    w, h, samples = 160, 120, 333
    ctx.upload_volume(vox)
    ctx.update_transfer_function(translucent, reset_state=True)
    ctx.set_camera(*cf.CAMERAS["h30_v20"])
    ctx.set_sampling(samples)
    ctx.set_mode(2)
    ref, st2 = ctx.render(w, h, stats=True)
    ctx.update_transfer_function(translucent, reset_state=True)     # re-arms the probe
    ctx.set_mode(1)
    first, st_first = ctx.render(w, h, stats=True)
    assert np.array_equal(first, ref)
    assert st_first["replay_rays"] > w * h / 100                    # the probe frame paid for the replay
    second, st = ctx.render(w, h, stats=True)
    assert np.array_equal(second, ref)
    assert st["replay_rays"] == 0 and st["long_rays"] == 0 and sum(st["rays_by_lane_taps"]) == 0
    assert st["samples_evaluated"] <= 1.1 * st2["samples_evaluated"]
    vol, _ = oracle.normalize(vox)
    lut = np.zeros((256, 4), np.float32)
    oracle.lut_update(lut, translucent)
    vp, m = oracle.camera(*cf.CAMERAS["h30_v20"], w, h)
    assert np.array_equal(second, oracle.render(vol, lut, vp[4:8], m, w, h, samples))
    # the benchmark table afterwards: the lane march runs again (only it counts rays by their taps)
    ctx.update_transfer_function(cf.TF_GRAPHS, reset_state=True)
    ctx.render(w, h)
    _, st = ctx.render(w, h, stats=True)
    assert sum(st["rays_by_lane_taps"]) == st["rays_hit"] > 0


def test_count_hit_rays(ctx, oracle):
    ctx.upload_volume(np.ones((4, 4, 4), np.uint8))
    for cam in cf.CAMERAS.values():
        ctx.set_camera(*cam)
        vp, m = oracle.camera(cam[0], cam[1], 200, 120)
        assert ctx.count_hit_rays(200, 120) == oracle.count_hits(vp[4:8], m, 200, 120)


def test_config2_strip_matches_the_checker(ctx, oracle):
    # BASELINE configs[1]: 512^3 uint16 phantom at 1920x1080, half-voxel sampling (1773 samples per ray):
    # a 48-row strip through the middle of the frame against the checker, in the fast and the skipping mode
    n, w, h = 512, 1920, 1080
    samples = int(math.ceil(2 * math.sqrt(3) * n)) - 1
    vox = oracle.phantom(n, np.uint16)
    ctx.upload_volume(vox)
    ctx.update_transfer_function(cf.TF_GRAPHS, reset_state=True)
    ctx.set_sampling(samples, (0, 0, 0, 0))
    cam = (math.radians(30.0), math.radians(20.0))
    ctx.set_camera(*cam)
    vol, _ = oracle.normalize(vox)
    del vox
    lut = np.zeros((256, 4), np.float32)
    oracle.lut_update(lut, cf.TF_GRAPHS)
    vp, m = oracle.camera(cam[0], cam[1], w, h)
    rows = (516, 564)
    want = oracle.render(vol, lut, vp[4:8], m, w, h, samples, rows=rows)[rows[0]:rows[1]]
    for mode in (1, 2):
        ctx.set_mode(mode)
        got = ctx.render(w, h, rect=(0, rows[0], w, rows[1]))
        check_frame(got, want, "512^3 1080p strip mode %d" % mode)


def test_full_size_volume_all_modes_agree(vr, torch):
    # BASELINE's headline size: 2048^3 uint16 (17 GB raw, 26 GB of bricks) generated on the device.
    # No CPU frame exists at this size; the property checked is that the fast march, the skipping
    # march and the plain reference-order march (already pinned to the checker at small sizes)
    # produce the same bytes, on a rectangle through the middle of a 4K frame.
    n, w, h = 2048, 3840, 2160
    samples = int(math.ceil(2 * math.sqrt(3) * n)) - 1
    with vr.VolumeRenderer(0) as c:
        raw = torch.empty(n ** 3, dtype=torch.int16, device="cuda")
        torch.cuda.synchronize()
        c.phantom_generate_device(raw.data_ptr(), 2, n)
        c.upload_volume_device(raw.data_ptr(), 2, n, n, n)
        c.synchronize()
        # spot-check the resident voxels against the generator's output
        probe = c.read_volume_normalized if False else None
        del probe
        sample = raw[:: 1000003].cpu().numpy().view(np.uint16)
        assert sample.max() > 60000 and (sample == 0).mean() > 0.5
        del raw
        torch.cuda.empty_cache()
        c.update_transfer_function(cf.TF_GRAPHS, reset_state=True)
        c.set_sampling(samples)
        c.set_camera(*cf.CAMERAS["h30_v20"])
        rect = (1800, 1000, 2056, 1128)
        frames = {}
        for mode in (0, 2, 1):
            c.set_mode(mode)
            frames[mode], st = c.render(w, h, stats=True, rect=rect)
            print("2048^3 mode %d: %s" % (mode, {k: st[k] for k in ("rays_hit", "samples_evaluated", "bricks_touched", "replay_rays")}))
            assert st["rays_hit"] == 256 * 128
        assert frames[0][..., :3].any()
        assert np.array_equal(frames[0], frames[2]) and np.array_equal(frames[0], frames[1])


def test_reference_data_and_defaults(ctx, vr, tmp_path):
    """The reference's own data and defaults through the HIP path: a block of its CThead slices (big-endian
    files, loadVolumeData's format) loaded by vr_load_volume_slices, rendered at its window size 1280x720
    (main.cpp:30) with samples = 200 (VolumeRendering.comp:173) and both sliders at 0 (main.cpp:69,77), in
    every mode, for both readings of the sampler's border colour and for the editor's default graphs."""
    sys_path = os.path.join(cf.GOLDEN)
    import sys
    sys.path.insert(0, sys_path)
    import make_cthead_crop as mk
    fx = np.load(os.path.join(cf.GOLDEN, "cthead_crop.npz"))
    mk.write_slices(fx["slices"], str(tmp_path))
    ctx.load_volume_data(str(tmp_path))
    info = ctx.volume_info()
    assert (info["nx"], info["ny"], info["nz"], info["dtype"]) == (88, 88, 113, 2) and info["max_value"] == 3272.0
    ctx.set_camera(*mk.CAMERA)
    w, h = mk.SIZE
    for name, border in mk.BORDERS.items():
        ctx.update_transfer_function(cf.TF_GRAPHS, reset_state=True)
        ctx.set_sampling(mk.SAMPLES, border)
        for mode in (0, 2, 1):
            ctx.set_mode(mode)
            check_frame(ctx.render(w, h), fx["frame_" + name], "CThead block, %s border, mode %d" % (name, mode))
    ctx.update_transfer_function(cf.lut_cases()["default_two_points"][0], reset_state=True)
    ctx.set_sampling(mk.SAMPLES, (0, 0, 0, 0))
    for mode in (0, 2, 1):
        ctx.set_mode(mode)
        got = ctx.render(w, h)
        assert np.array_equal(got, fx["frame_default_graphs"]) and not got[..., :3].any()
    # the frame that allocates nothing takes the same path
    import torch
    buf = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.update_transfer_function(cf.TF_GRAPHS, reset_state=True)
    ctx.render_static(w, h, buf.data_ptr())
    ctx.synchronize()
    assert np.array_equal(buf.cpu().numpy(), fx["frame_transparent"])


silhouette_rects = cf.silhouette_rects


def test_full_size_volume_silhouette_and_whole_frames(vr, torch):
    """The headline size where it is hard: 2048^3 uint16 at 3840x2160, three cameras of the orbit.  Whole frames
    of the fast march against the skipping reference-order march (all 8.3 M pixels), and rectangles on the
    object's outline against the literal shader loop; the fast march must have used its hand-over and its
    replay on them."""
    n, w, h = 2048, 3840, 2160
    samples = int(math.ceil(2 * math.sqrt(3) * n)) - 1
    with vr.VolumeRenderer(0) as c:
        raw = torch.empty(n ** 3, dtype=torch.int16, device="cuda")
        torch.cuda.synchronize()
        c.phantom_generate_device(raw.data_ptr(), 2, n)
        c.upload_volume_device(raw.data_ptr(), 2, n, n, n)
        c.synchronize()
        del raw
        torch.cuda.empty_cache()
        c.update_transfer_function(cf.TF_GRAPHS, reset_state=True)
        c.set_sampling(samples)
        for deg in (-180.0, -75.0, 70.0):
            c.set_camera(math.radians(deg), math.radians(20.0))
            c.set_mode(1)
            fast, st = c.render(w, h, stats=True)
            assert st["long_rays"] > 0 and st["replay_rays"] > 0 and st["long_samples"] > 0
            c.set_mode(2)
            skip = c.render(w, h)
            assert np.array_equal(fast, skip), (deg, cf.frame_report(fast, skip))
            rects = silhouette_rects(fast)
            assert len(rects) >= 3
            handed = replayed = 0
            for rect in rects:
                x0, y0, x1, y1 = rect
                crop = fast[y0:y1, x0:x1]
                assert crop[..., :3].any() and not crop[..., :3].any(axis=-1).all(), rect     # object and background
                c.set_mode(0)
                literal = c.render(w, h, rect=rect)
                assert np.array_equal(literal, crop), (deg, rect, cf.frame_report(literal, crop))
                c.set_mode(1)
                again, rs = c.render(w, h, stats=True, rect=rect)
                assert np.array_equal(again, crop), (deg, rect)
                handed += rs["long_rays"]
                replayed += rs["replay_rays"]
            print("camera %.0f: %d rectangles on the outline, %d rays handed over, %d replayed" % (deg, len(rects), handed, replayed))
            assert handed > 0 and replayed > 0


def test_config3_bands_reassemble(vr, torch):
    """BASELINE configs[2]: 1024^3 uint16 at 3840x2160, sort-first.  The bands of 2, 4 and 8 parts (fast march)
    put together give the frame the skipping reference-order march renders in one piece, and rectangles on the
    outline equal the literal shader loop."""
    n, w, h = 1024, 3840, 2160
    samples = int(math.ceil(2 * math.sqrt(3) * n)) - 1
    with vr.VolumeRenderer(0) as c:
        raw = torch.empty(n ** 3, dtype=torch.int16, device="cuda")
        torch.cuda.synchronize()
        c.phantom_generate_device(raw.data_ptr(), 2, n)
        c.upload_volume_device(raw.data_ptr(), 2, n, n, n)
        c.synchronize()
        del raw
        c.update_transfer_function(cf.TF_GRAPHS, reset_state=True)
        c.set_sampling(samples)
        c.set_camera(math.radians(-140.0), math.radians(20.0))
        c.set_mode(2)
        whole = c.render(w, h)
        assert whole[..., :3].any()
        c.set_mode(0)
        for rect in silhouette_rects(whole, size=96):
            x0, y0, x1, y1 = rect
            assert np.array_equal(c.render(w, h, rect=rect), whole[y0:y1, x0:x1]), rect
        c.set_mode(1)
        for parts in (2, 4, 8):
            got = np.zeros_like(whole)
            for part in range(parts):
                got[vr.band_row_indices(h, part, parts)] = c.render_bands(w, h, part, parts)
            assert np.array_equal(got, whole), (parts, cf.frame_report(got, whole))
        # frames in flight at this size (the finishing kernels of a frame run for as long as the next march):
        # six frames of an orbit back to back into two buffers, nobody waits; the buffers end up with the last
        # two, and those equal the frames rendered one at a time (tiles longest first) and in mode 2
        cams = [(math.radians(-140.0 + k), math.radians(20.0)) for k in range(6)]
        alone = []
        for cam in cams[-2:]:
            c.set_camera(*cam)
            c.set_mode(2)
            ref = c.render(w, h)
            c.set_mode(1)
            one = c.render(w, h)
            assert np.array_equal(one, ref)
            alone.append(one)
        bufs = [torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
        torch.cuda.synchronize()
        c.set_pipeline(2)
        for k, cam in enumerate(cams):
            c.set_camera(*cam)
            c.render_device(w, h, bufs[k & 1].data_ptr())
        c.synchronize()
        c.set_pipeline(1)
        assert np.array_equal(bufs[0].cpu().numpy(), alone[0]) and np.array_equal(bufs[1].cpu().numpy(), alone[1])


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_sort_first_bands_tile_the_frame(ctx, vr, oracle, mode):
    w, h = 70, 53                                   # ragged: last band has 5 rows, last tile column 6 pixels
    ctx.upload_volume(oracle.phantom(32, np.uint16))
    ctx.update_transfer_function(cf.TF_GRAPHS, reset_state=True)
    ctx.set_camera(*cf.CAMERAS["h30_v20"])
    ctx.set_sampling(90)
    ctx.set_mode(mode)
    full = ctx.render(w, h)
    for parts in (1, 2, 3, 5):
        got = np.zeros_like(full)
        for part in range(parts):
            bands = ctx.render_bands(w, h, part, parts)
            rows = vr.band_row_indices(h, part, parts)
            assert bands.shape == (rows.size, w, 4)
            got[rows] = bands
        assert np.array_equal(got, full), parts


def test_tiles_in_order_of_their_time(ctx, vr, oracle):
    """Frames rendered one at a time launch their tiles longest first, by the times the previous frame over
    the same tiles noted (launches of 8192 tiles and more).  Any order gives the same bytes: a sequence of
    views, a change of size in between, and the bands of a split, against the reference-order march with
    skipping of the same context."""
    ctx.upload_volume(oracle.phantom(64, np.uint16))
    ctx.update_transfer_function(cf.TF_GRAPHS, reset_state=True)
    ctx.set_sampling(150)
    cams = [(math.radians(20.0 + 7.0 * k), math.radians(15.0 - 3.0 * k)) for k in range(4)]

    def reference(w, h, cam):
        ctx.set_mode(2)
        ctx.set_camera(*cam)
        img = ctx.render(w, h)
        ctx.set_mode(1)
        return img

    ctx.set_mode(1)
    for w, h in ((1024, 520), (1100, 500), (1024, 520)):      # 8320, 8694 tiles; the first size again
        for cam in cams:
            ctx.set_camera(*cam)
            got = ctx.render(w, h)
            assert np.array_equal(got, reference(w, h, cam)), (w, h, cam)
    # one part of a split: its own tiles, its own lists (2048 x 1040: 16640 tiles a part)
    w, h = 2048, 1040
    for cam in cams[:3]:
        ctx.set_camera(*cam)
        full = reference(w, h, cam)
        for part in (1, 0, 1):
            ctx.set_camera(*cam)
            bands = ctx.render_bands(w, h, part, 2)
            assert np.array_equal(bands, full[vr.band_row_indices(h, part, 2)]), (cam, part)


# ---- histogram (prepareHistogramVolumeData, main.cpp:166-181) -----------------------------------

def test_tune_keeps_the_frame(ctx, oracle):
    """vr_tune picks the per-lane work budget by timing; the frame does not depend on it."""
    ctx.upload_volume(oracle.phantom(64, np.uint16))
    ctx.update_transfer_function(cf.TF_GRAPHS, reset_state=True)
    ctx.set_camera(*cf.CAMERAS["h30_v20"])
    ctx.set_sampling(250)
    ctx.set_mode(0)
    ref = ctx.render(160, 120)
    candidates = (8, 16, 32, 48, 64, 96, 128, 160)
    assert ctx.tune(160, 120) in candidates
    assert np.array_equal(ctx.render(160, 120), ref)          # tune leaves mode and pipeline as they were
    ctx.set_mode(1)
    assert np.array_equal(ctx.render(160, 120), ref)
    assert ctx.tune(160, 120, 1, 3) in candidates
    assert np.array_equal(ctx.render(160, 120), ref)
    # a part that owns no band (more parts than bands) has nothing to measure: the budget stays
    kept = ctx.tune(160, 120, 1, 3)
    assert ctx.tune(160, 16, 5, 8) == kept


def test_assemble_gathered_bands(ctx, oracle, torch):
    """vr_assemble_bands: the parts' band buffers, laid out as a gather leaves them, give the whole frame."""
    import vulkanrenderer_amd as vr
    ctx.upload_volume(oracle.phantom(48, np.uint16))
    ctx.update_transfer_function(cf.TF_GRAPHS, reset_state=True)
    ctx.set_camera(*cf.CAMERAS["h30_v20"])
    ctx.set_sampling(120)
    ctx.set_mode(1)
    w, h = 150, 117           # 8 bands, the last one short
    whole = ctx.render(w, h)
    for parts in (1, 3, 8, 11):
        stride = vr.band_rows(h, 0, parts) * w * 4 + 64       # some padding between the parts
        gathered = torch.zeros(parts * stride, dtype=torch.uint8, device="cuda")
        frame = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        for p in range(parts):
            if vr.band_rows(h, p, parts) > 0:
                ctx.render_bands(w, h, p, parts, ptr=gathered.data_ptr() + p * stride)
        ctx.assemble_bands(w, h, parts, gathered.data_ptr(), stride, frame.data_ptr())
        ctx.synchronize()
        assert np.array_equal(frame.cpu().numpy(), whole), parts
    with pytest.raises(vr.VrError):
        ctx.assemble_bands(w, h, 3, gathered.data_ptr(), 16, frame.data_ptr())


@pytest.mark.parametrize("dtype,shape,top,given_max", [
    (np.uint8, (20, 33, 41), 255, 0.0), (np.uint16, (17, 24, 30), 65535, 0.0), (np.uint16, (1, 1, 1), 65535, 0.0),
    (np.uint16, (16, 16, 16), 65535, 0.0),        # the last brick of every axis holds two layers of voxels
    (np.uint16, (6, 14, 130), 4000, 4095.0),      # 12-bit data with the maximum given; x ends exactly with a brick's apron
    (np.uint8, (9, 70, 8), 7, 0.0),               # few values: most bins are empty, some values fall between bins
])
def test_histogram_bit_exact(ctx, oracle, dtype, shape, top, given_max):
    rng = np.random.default_rng(21)
    vox = (rng.random(shape) ** 3 * top).astype(dtype)          # skewed: many small values, zeros included
    vox[0, 0, 0] = top                                           # not all zero, and the value range is known
    ctx.upload_volume(vox, given_max)
    want = oracle.histogram(oracle.normalize(vox, given_max)[0])
    got = ctx.prepare_histogram_volume_data()
    assert cf.same_bits(got, want)
    # the fullest bin is 1; a volume whose every voxel falls between two bins divides 0 by 0 as the reference would
    assert got.max() == 1.0 or np.isnan(got).all()


def test_histogram_of_the_phantom(ctx, oracle):
    vox = oracle.phantom(96, np.uint16)
    ctx.upload_volume(vox)
    got = ctx.prepare_histogram_volume_data()
    assert cf.same_bits(got, oracle.histogram(oracle.normalize(vox)[0]))
    assert got.max() == 1.0 and got[0] == 1.0                    # the exterior zeros dominate


# ---- the C++ host program on the C ABI ------------------------------------------------------------

@pytest.mark.parametrize("in_flight", [1, 2])
def test_mapped_frames_for_a_host_consumer(ctx, vr, oracle, in_flight):
    """vr_render_mapped / vr_map_frame: an orbit whose frames are copied into the context's page-locked ring while
    the next frame renders; every frame equals the one vr_render returns, in both fast and reference-order mode,
    with one and with two frames in flight, across a change of image size; a slot that is still mapped when
    its turn comes again is refused."""
    ctx.upload_volume(oracle.phantom(56, np.uint16))
    ctx.update_transfer_function(cf.TF_GRAPHS, reset_state=True)
    ctx.set_sampling(160)
    cams = [(math.radians(-60.0 + 25.0 * k), math.radians(10.0 + 4.0 * k)) for k in range(7)]
    for mode, (w, h) in ((1, (200, 144)), (0, (120, 72)), (1, (333, 211))):
        ctx.set_mode(mode)
        ctx.set_pipeline(1)
        want = []
        for cam in cams:
            ctx.set_camera(*cam)
            want.append(ctx.render(w, h))
        ctx.set_pipeline(in_flight)
        previous = None
        for k in range(len(cams) + 1):
            slot = None
            if k < len(cams):
                ctx.set_camera(*cams[k])
                slot = ctx.render_mapped(w, h)
            if previous is not None:
                got = ctx.map_frame(previous, w, h)
                assert np.array_equal(got, want[k - 1]), (mode, k - 1, cf.frame_report(got, want[k - 1]))
                ctx.unmap_frame(previous)
            previous = slot
    ctx.set_pipeline(1)
    # three slots: with one left mapped, the third render after it has nowhere to go
    ctx.set_mode(1)
    first = ctx.render_mapped(64, 64)
    ctx.map_frame(first, 64, 64)
    ctx.render_mapped(64, 64)
    ctx.render_mapped(64, 64)
    with pytest.raises(vr.VrError, match="mapped"):
        ctx.render_mapped(64, 64)
    ctx.unmap_frame(first)
    ctx.unmap_frame(ctx.render_mapped(64, 64))
    with pytest.raises(vr.VrError):
        ctx.map_frame(7, 64, 64)


@pytest.mark.parametrize("hand_off", ["mapped", "pageable"])
def test_orbit_example_writes_the_checkers_frames(oracle, tmp_path, hand_off):
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "orbit")
    if not os.path.exists(exe):
        pytest.fail("examples/orbit is not built: make -C vulkanrenderer_amd/csrc")
    w, h, n, samples, frames = 96, 80, 48, 120, 3
    res = subprocess.run([exe, "--phantom", str(n), "--size", "%dx%d" % (w, h), "--frames", str(frames), "--samples", str(samples),
                          "--start", "10", "--step", "40", "--elevation", "25", "--out", str(tmp_path)] +
                         (["--pageable"] if hand_off == "pageable" else []), capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    vol, _ = oracle.normalize(oracle.phantom(n, np.uint16))
    lut = np.zeros((256, 4), np.float32)
    oracle.lut_update(lut, cf.TF_GRAPHS)
    rad = np.float32(0.01745329251994329576923690768489)
    for f in range(frames):
        data = open(os.path.join(str(tmp_path), "frame_%04d.ppm" % f), "rb").read()
        header = b"P6\n%d %d\n255\n" % (w, h)
        assert data.startswith(header)
        got = np.frombuffer(data[len(header):], np.uint8).reshape(h, w, 3)
        hdeg = np.float32(10.0) + np.float32(40.0) * np.float32(f)
        vp, m = oracle.camera(float(hdeg * rad), float(np.float32(25.0) * rad), w, h)
        want = oracle.render(vol, lut, vp[4:8], m, w, h, samples)
        assert np.array_equal(got, want[..., :3]), f


# ---- the shader's commented-out lighting (VolumeRendering.comp:109-152,195-198) ---------------------
# The device's powf and the host's may differ in the last bit, so a byte may now and then differ by
# one from the checker's: the frame bar (<= 1/255, >= 99.9 % exact) applies, not byte equality.  The
# three modes run the same device code per tap and must agree byte for byte among themselves.

@pytest.mark.parametrize("sname", list(cf.SCENES))
def test_shaded_golden_frames(ctx, oracle, golden, sname):
    ctx.upload_volume(oracle.phantom(32, np.uint8))
    ctx.update_transfer_function(cf.TF_GRAPHS, reset_state=True)
    ctx.set_sampling(200, (0, 0, 0, 0))
    ctx.set_shading(True, cf.SCENES[sname])
    for cname, cam in cf.CAMERAS.items():
        ctx.set_camera(*cam)
        frames = []
        for mode in (0, 2, 1):
            ctx.set_mode(mode)
            frames.append(ctx.render(64, 64))
            check_frame(frames[-1], golden["shaded_u8_32_%s_%s" % (cname, sname)], "shaded %s %s mode %d" % (cname, sname, mode),
                        exact_required=False)
        assert np.array_equal(frames[0], frames[1]) and np.array_equal(frames[0], frames[2]), (cname, sname)
    # switching it off again gives the plain frame
    ctx.set_shading(False)
    ctx.set_mode(1)
    check_frame(ctx.render(64, 64), golden["frame_u8_32_" + cname], "shading off again")


def test_shaded_frames_larger_volume(ctx, oracle):
    vox = oracle.phantom(96, np.uint16)
    w, h, samples = 160, 120, 333
    ctx.upload_volume(vox)
    ctx.update_transfer_function(cf.TF_GRAPHS, reset_state=True)
    ctx.set_sampling(samples, (0, 0, 0, 0))
    ctx.set_shading(True)          # the reference's constants
    vol, _ = oracle.normalize(vox)
    lut = np.zeros((256, 4), np.float32)
    oracle.lut_update(lut, cf.TF_GRAPHS)
    for cam in (cf.CAMERAS["h30_v20"], cf.CAMERAS["hm135_vm60"]):
        ctx.set_camera(*cam)
        vp, m = oracle.camera(cam[0], cam[1], w, h)
        want = oracle.render(vol, lut, vp[4:8], m, w, h, samples, scene=cf.SCENES["reference"])
        ctx.set_mode(0)
        ref = ctx.render(w, h)
        check_frame(ref, want, "shaded 96^3 reference order", exact_required=False)
        ctx.set_mode(2)
        assert np.array_equal(ctx.render(w, h), ref)
        ctx.set_mode(1)
        fast, st = ctx.render(w, h, stats=True)
        assert np.array_equal(fast, ref)
        assert st["samples_evaluated"] < st["samples_nominal"]
        # bands and two frames in flight take the same path
        parts = [ctx.render_bands(w, h, p, 3) for p in range(3)]
        assert sum(x.shape[0] for x in parts) == h


def test_shading_arguments(vr, ctx, oracle):
    ctx.upload_volume(oracle.phantom(16, np.uint8))
    for bad in ((-0.1, 0, 0, 10), (0, 1.5, 0, 10), (0, 0, float("nan"), 10), (0, 0, 0, 0.5), (0, 0, 0, 200.0)):
        with pytest.raises(vr.VrError):
            ctx.set_shading(True, bad)
    ctx.set_shading(True, (0.0, 0.0, 0.0, 1.0))
    # all weights zero: the term adds 0 to every tap
    ctx.update_transfer_function(cf.TF_GRAPHS, reset_state=True)
    ctx.set_camera(*cf.CAMERAS["h30_v20"])
    ctx.set_sampling(40)
    ctx.set_mode(1)
    zero = ctx.render(40, 40)
    ctx.set_shading(False)
    assert np.array_equal(zero, ctx.render(40, 40))
    # the partial march (sort-last step 1) has no lighting: a whole-volume context with lighting on refuses it
    ctx.set_shading(True)
    with pytest.raises(vr.VrError, match="lighting"):
        ctx.render_partial(40, 40)
    ctx.set_shading(False)
    assert ctx.render_partial(40, 40).shape == (1600, 5)
    # a context holding a box of the volume refuses the lighting, and the other way round
    from vulkanrenderer_amd import sortlast
    vox = oracle.phantom(32, np.uint8)
    ctx.set_shading(True)
    with pytest.raises(vr.VrError):
        sortlast.upload_box_of(ctx, vox, (2, 1, 1), (0, 0, 0), "cuda")
    ctx.set_shading(False)
    sortlast.upload_box_of(ctx, vox, (2, 1, 1), (0, 0, 0), "cuda")
    with pytest.raises(vr.VrError):
        ctx.set_shading(True)


def _run_bench(argv, timeout=240):
    """`python bench.py ...` as the driver starts it; with --gpus N > 1 it launches its own ranks.  Rehearsal
    switches: the ranks share this one GPU and exchange over gloo through host memory."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VR_BENCH_BACKEND="gloo", VR_BENCH_ONE_DEVICE="1", VR_BENCH_WATCHDOG="150")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + argv, capture_output=True, text=True, timeout=timeout, env=env, cwd=root)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_launches_its_own_ranks_sort_first():
    """bench.py --gpus 2: band split, two band buffers in turn, gather, assembly, max over ranks; a small workload."""
    line = _run_bench(["--gpus", "2", "--steps", "6", "--warmup", "2", "--config", "c3", "--volume", "128", "--width", "320", "--height", "200",
                       "--stats-frames", "2"])
    assert line["n_gpus"] == 2 and line["steps"] == 6 and line["value"] > 0 and line["unit"] == "Gsamples/s"
    assert line["scaling"] == "strong" and line["config"]["name"] == "c3" and "overridden" in line["config"]["workload"]
    assert line["roofline"]["frac"] > 0 and line["roofline"]["bound"] == "hbm"


@pytest.mark.parametrize("gpus,schedule", [(8, "binary_swap"), (4, "direct_send")])
def test_bench_sort_last_configuration(gpus, schedule):
    """bench.py --config c5: one box per rank, exchange, decision, re-march, gather; per-stage times in the line."""
    line = _run_bench(["--gpus", str(gpus), "--config", "c5", "--schedule", schedule, "--steps", "3", "--warmup", "1", "--volume", "96",
                       "--width", "160", "--height", "120", "--stats-frames", "2"])
    assert line["n_gpus"] == gpus and line["scaling"] == "weak" and line["value"] > 0
    assert set(line["stages"]) == {"partial_march", "exchange_merge", "decide", "remarch", "gather"}
    assert len(line["partial_march_ms_per_rank"]) == gpus and line["partial_march_max_over_mean"] >= 1.0
    assert schedule.replace("_", " ") in line["config"]["workload"]


def test_bench_single_gpu_configurations():
    for name, extra in (("c1", []), ("c2", ["--volume", "128", "--width", "320", "--height", "180"]), ("c5", ["--volume", "96", "--width", "160", "--height", "120"])):
        line = _run_bench(["--config", name, "--steps", "4", "--warmup", "1", "--stats-frames", "2", "--no-cpu-baseline"] + extra)
        assert line["n_gpus"] == 1 and line["config"]["name"] == name and line["value"] > 0 and line["ms_per_step"] > 0
        assert "roofline" in line
        if name == "c1":
            assert line["config"]["volume"] == [256, 256, 256] and line["config"]["samples_per_ray"] == 200
            assert "4 frames" not in line["config"]["workload"] and "fixed camera" in line["config"]["workload"]
            assert line["roofline"]["frac_whole_frame"] > 0


def test_two_frames_in_flight_render_the_same_frames(ctx, oracle, torch):
    """vr_set_pipeline(2): the finishing kernels of a frame overlap the next frame's march; every frame,
    each in a buffer of its own or in two buffers used in turn, equals the frame rendered alone."""
    n, w, h, samples = 96, 200, 150, 300
    vox = oracle.phantom(n, np.uint16)
    ctx.upload_volume(vox)
    ctx.update_transfer_function(cf.TF_GRAPHS, reset_state=True)
    ctx.set_sampling(samples)
    ctx.set_mode(1)
    cams = [(math.radians(-170.0 + 37.0 * k), math.radians(20.0 - 9.0 * k)) for k in range(7)]
    alone = []
    for cam in cams:
        ctx.set_camera(*cam)
        alone.append(ctx.render(w, h))
    ctx.set_pipeline(2)
    bufs = [torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda") for _ in cams]
    torch.cuda.synchronize()
    for cam, b in zip(cams, bufs):
        ctx.set_camera(*cam)
        ctx.render_device(w, h, b.data_ptr())
    ctx.synchronize()
    for k, b in enumerate(bufs):
        assert np.array_equal(b.cpu().numpy(), alone[k]), k
    # two buffers in turn, each frame read back through a stream that waits for it
    pair = [torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    got = []
    for k, cam in enumerate(cams):
        ctx.set_camera(*cam)
        ctx.render_device(w, h, pair[k & 1].data_ptr())
        if k >= 1:
            # frame k - 1 sits in the other buffer; its finishing kernels may still be running
            pass
        ctx.wait_frame(torch.cuda.current_stream().cuda_stream)
        got.append(pair[k & 1].cpu().numpy().copy())      # .cpu() synchronises torch's stream: the buffer is free again
    for k in range(len(cams)):
        assert np.array_equal(got[k], alone[k]), k
    # no waits by the caller at all: a frame that writes where a frame still being finished writes waits
    # for it by itself, so one buffer ends up holding the last frame and two buffers the last two
    for count in (1, 2):
        for k, cam in enumerate(cams):
            ctx.set_camera(*cam)
            ctx.render_device(w, h, pair[k % count].data_ptr())
        ctx.synchronize()
        for k in range(len(cams) - count, len(cams)):
            assert np.array_equal(pair[k % count].cpu().numpy(), alone[k]), (count, k)
    # a transfer-function change between pipelined frames is ordered after the frames in flight
    ctx.set_camera(*cams[0])
    ctx.render_device(w, h, pair[0].data_ptr())
    ctx.update_transfer_function([cf.TF_GRAPHS[0], cf.TF_GRAPHS[1], cf.TF_GRAPHS[2], [(0, 0.0), (200, 0.0), (483, 0.9)]])
    ctx.render_device(w, h, pair[1].data_ptr())
    ctx.synchronize()
    assert np.array_equal(pair[0].cpu().numpy(), alone[0])
    ctx.set_pipeline(1)
    assert np.array_equal(ctx.render(w, h), pair[1].cpu().numpy())
    assert not np.array_equal(pair[1].cpu().numpy(), alone[0])


def test_eight_pipelined_contexts_in_one_process(vr, oracle, torch):
    """Eight contexts with two frames in flight each: sixteen streams of this library in one process, more
    than the four hardware queues a process gets (GPU_MAX_HW_QUEUES), so the runtime multiplexes them.
    Every context renders frames in turn, interleaved with the others; all frames equal the one rendered
    alone; everything is destroyed again."""
    n, w, h, samples = 64, 160, 120, 200
    vox = oracle.phantom(n, np.uint16)
    cams = [(math.radians(-150.0 + 40.0 * k), math.radians(15.0)) for k in range(3)]
    with vr.VolumeRenderer(0) as c:
        c.upload_volume(vox)
        c.update_transfer_function(cf.TF_GRAPHS, reset_state=True)
        c.set_sampling(samples)
        c.set_mode(1)
        alone = []
        for cam in cams:
            c.set_camera(*cam)
            alone.append(c.render(w, h))
    ctxs, bufs = [], []
    try:
        for k in range(8):
            c = vr.VolumeRenderer(0)
            c.upload_volume(vox)
            c.update_transfer_function(cf.TF_GRAPHS, reset_state=True)
            c.set_sampling(samples)
            c.set_mode(1)
            c.set_pipeline(2)
            ctxs.append(c)
            bufs.append([torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda") for _ in cams])
        torch.cuda.synchronize()
        for f, cam in enumerate(cams):          # frame f of every context before frame f + 1 of any
            for c, b in zip(ctxs, bufs):
                c.set_camera(*cam)
                c.render_device(w, h, b[f].data_ptr())
        for c in ctxs:
            c.synchronize()
        for k, b in enumerate(bufs):
            for f in range(len(cams)):
                assert np.array_equal(b[f].cpu().numpy(), alone[f]), (k, f)
    finally:
        for c in ctxs:
            c.close()


# ---- the table set every frame, as the reference's frame loop does (main.cpp:800-804, 687-736) ---------------

def test_unchanged_table_call_costs_nothing_and_changes_nothing(ctx, oracle, torch):
    """renderCommands rebuilds and uploads the table every frame whether or not a control point moved.  The same
    graphs again leave every bit of the table as it is, so the call must neither touch the table nor disturb the
    frames in flight; with reset_state it still zeroes what the graphs do not cover when the table is not already
    what they make of a zeroed one."""
    vox = oracle.phantom(64, np.uint16)
    ctx.upload_volume(vox)
    frac = [[(100.0, 0.2), (300.0, 0.8)]] * 3 + [[(150.0, 0.0), (300.0, 0.7)]]    # entries below the first point are not covered
    ctx.update_transfer_function(cf.TF_GRAPHS, reset_state=True)
    ctx.update_transfer_function(frac, reset_state=False)             # stale entries of the first table survive
    lut = np.zeros((256, 4), np.float32)
    oracle.lut_update(lut, cf.TF_GRAPHS)
    oracle.lut_update(lut, frac)
    assert cf.same_bits(ctx.transfer_function(), lut)
    ctx.update_transfer_function(frac, reset_state=False)             # same graphs: nothing changes
    assert cf.same_bits(ctx.transfer_function(), lut)
    ctx.update_transfer_function(frac, reset_state=True)              # same graphs, but now from a zeroed table
    lut0 = np.zeros((256, 4), np.float32)
    oracle.lut_update(lut0, frac)
    assert cf.same_bits(ctx.transfer_function(), lut0) and not cf.same_bits(lut0, lut)
    ctx.update_transfer_function(frac, reset_state=True)
    ctx.update_transfer_function(frac, reset_state=False)
    assert cf.same_bits(ctx.transfer_function(), lut0)
    # pipelined frames with the call in front of each: same frames as without it
    w, h, samples = 200, 150, 220
    ctx.update_transfer_function(cf.TF_GRAPHS, reset_state=True)
    ctx.set_sampling(samples, (0, 0, 0, 0))
    ctx.set_mode(1)
    cams = [(math.radians(-60.0 + 7 * k), math.radians(15.0)) for k in range(6)]
    want = []
    for cam in cams:
        ctx.set_camera(*cam)
        want.append(ctx.render(w, h))
    bufs = [torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda") for _ in cams]
    torch.cuda.synchronize()
    ctx.set_pipeline(2)
    for cam, buf in zip(cams, bufs):
        ctx.update_transfer_function(cf.TF_GRAPHS, reset_state=False)
        ctx.set_camera(*cam)
        ctx.render_device(w, h, buf.data_ptr())
    ctx.synchronize()
    ctx.set_pipeline(1)
    for k in range(len(cams)):
        assert np.array_equal(bufs[k].cpu().numpy(), want[k]), k


def _moving_tables():
    """Tables of an edited control graph: the alpha threshold moves up and down (the set of transparent entries
    changes both ways), a colour changes alone (it does not), alpha moves between non-zero values (it does not)."""
    xs = (0.0, 120.0, 300.0, 483.0)
    tables = []
    for thr in (120.0, 160.0, 90.0, 40.0, 200.0, 120.0):
        ax = (0.0, thr, 300.0, 483.0)
        tables.append([list(zip(xs, (0, .9, 1, 1))), list(zip(xs, (0, .3, .8, 1))), list(zip(xs, (0, .1, .5, 1))),
                       list(zip(ax, (0, 0, .05, .6)))])
    tables.append([list(zip(xs, (0, .2, .4, 1))), list(zip(xs, (0, .9, .8, 1))), list(zip(xs, (0, .1, .9, .3))), list(zip(xs, (0, 0, .05, .6)))])
    tables.append([list(zip(xs, (0, .2, .4, 1))), list(zip(xs, (0, .9, .8, 1))), list(zip(xs, (0, .1, .9, .3))), list(zip(xs, (0, 0, .3, .9)))])
    return tables


def test_table_changed_between_frames(ctx, oracle, torch):
    """A control point dragged from frame to frame: every frame equals the checker's frame for its table, in the
    fast mode and in the skipping mode (both rely on occupancy tables rebuilt on the GPU after each change, only
    for the bricks the change concerns), one frame at a time and with frames in flight."""
    vox = oracle.phantom(80, np.uint16)
    vol, _ = oracle.normalize(vox)
    w, h, samples = 160, 120, 277
    ctx.upload_volume(vox)
    ctx.set_sampling(samples, (0, 0, 0, 0))
    cam = cf.CAMERAS["h30_v20"]
    ctx.set_camera(*cam)
    vp, m = oracle.camera(cam[0], cam[1], w, h)
    tables = _moving_tables()
    lut = np.zeros((256, 4), np.float32)
    want = []
    ctx.update_transfer_function(tables[0], reset_state=True)
    for k, graphs in enumerate(tables):
        oracle.lut_update(lut, graphs)
        want.append(oracle.render(vol, lut, vp[4:8], m, w, h, samples))
        ctx.update_transfer_function(graphs, reset_state=False)
        for mode in (1, 2, 0):
            ctx.set_mode(mode)
            got = ctx.render(w, h)
            assert np.array_equal(got, want[k]), (k, mode, cf.frame_report(got, want[k]))
    assert any(not np.array_equal(want[k], want[k + 1]) for k in range(len(want) - 1))
    # the same walk with frames in flight, the table set in front of every frame, nobody waits in between
    ctx.update_transfer_function(tables[0], reset_state=True)
    bufs = [torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda") for _ in tables]
    torch.cuda.synchronize()
    ctx.set_mode(1)
    ctx.set_pipeline(2)
    for graphs, buf in zip(tables, bufs):
        ctx.update_transfer_function(graphs, reset_state=False)
        ctx.render_device(w, h, buf.data_ptr())
    ctx.synchronize()
    ctx.set_pipeline(1)
    for k in range(len(tables)):
        got = bufs[k].cpu().numpy()
        assert np.array_equal(got, want[k]), (k, cf.frame_report(got, want[k]))
    # border colour and lighting change what the decisions rely on without a new table
    ctx.set_sampling(samples, (0.1, 0.2, 0.3, 0.5))
    got = ctx.render(w, h)
    assert np.array_equal(got, oracle.render(vol, lut, vp[4:8], m, w, h, samples, border=(0.1, 0.2, 0.3, 0.5)))
    ctx.set_sampling(samples, (0, 0, 0, 0))
    assert np.array_equal(ctx.render(w, h), want[-1])


def test_tables_in_turn_while_frames_are_in_flight(ctx, oracle, torch):
    """Colours and opacities edited in front of every frame while the set of transparent entries stays: such a table
    goes into the next of the library's table slots and the frames in flight finish on theirs.  Every frame of a
    run without waits equals the checker's frame for its own table -- through more changes than there are slots,
    several changes between two frames, a change that starts from a zeroed table, and a change of the transparent
    set in the middle -- and the table read back at the end is the checker's bit for bit."""
    vox = oracle.phantom(80, np.uint16)
    vol, _ = oracle.normalize(vox)
    w, h, samples = 160, 120, 277
    ctx.upload_volume(vox)
    ctx.set_sampling(samples, (0, 0, 0, 0))
    cam = cf.CAMERAS["h30_v20"]
    ctx.set_camera(*cam)
    vp, m = oracle.camera(cam[0], cam[1], w, h)
    xs = (0.0, 120.0, 300.0, 483.0)

    def graphs(k, thr=120.0):
        a = 0.05 + 0.03 * (k % 5)
        return [list(zip(xs, (0, .9 - .05 * (k % 7), 1, 1))), list(zip(xs, (0, .3, .8 - .04 * (k % 3), 1))),
                list(zip(xs, (0, .1 + .02 * k, .5, 1))), list(zip((0.0, thr, 300.0, 483.0), (0, 0, a, .6 + .02 * (k % 4))))]

    # (graphs, reset_state, frame rendered behind the call)
    steps = [(graphs(k), False, True) for k in range(7)]
    steps += [(graphs(10 + k), False, k == 5) for k in range(6)]           # six calls, one frame
    steps += [(graphs(20), True, True)]                                    # from a zeroed table, same set
    steps += [(graphs(21, thr=170.0), False, True)]                        # the transparent range moves: tables rebuilt
    steps += [(graphs(22 + k, thr=170.0), False, True) for k in range(5)]  # and slots in turn again behind it
    lut = np.zeros((256, 4), np.float32)
    ctx.update_transfer_function(graphs(0), reset_state=True)
    oracle.lut_update(lut, graphs(0))
    ctx.set_mode(1)
    ctx.render(w, h)                                                       # occupancy tables built for this set
    ctx.set_pipeline(2)
    want, bufs = [], []
    for g, reset, framed in steps:
        if framed:
            bufs.append(torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda"))
    torch.cuda.synchronize()
    at = 0
    for g, reset, framed in steps:
        if reset:
            lut[:] = 0
        oracle.lut_update(lut, g)
        ctx.update_transfer_function(g, reset_state=reset)
        if framed:
            want.append(oracle.render(vol, lut, vp[4:8], m, w, h, samples))
            ctx.render_device(w, h, bufs[at].data_ptr())
            at += 1
    ctx.synchronize()
    ctx.set_pipeline(1)
    for k in range(len(want)):
        got = bufs[k].cpu().numpy()
        assert np.array_equal(got, want[k]), (k, cf.frame_report(got, want[k]))
    assert sum(not np.array_equal(want[k], want[k + 1]) for k in range(len(want) - 1)) >= len(want) - 2
    assert cf.same_bits(ctx.transfer_function(), lut)
    for mode in (2, 0):                                                    # the other marches read the current slot too
        ctx.set_mode(mode)
        assert np.array_equal(ctx.render(w, h), want[-1]), mode


def test_many_control_points_take_the_staged_path(ctx, oracle):
    """More points than travel with the launch (256): the graphs go through the staging buffer; same table."""
    xs = np.linspace(0.0, 480.0, 300)
    graphs = [[(float(x), float((k * 37 + i * 11) % 101) / 100.0) for k, x in enumerate(xs)] for i in range(4)]
    ctx.update_transfer_function(graphs, reset_state=True)
    lut = np.zeros((256, 4), np.float32)
    oracle.lut_update(lut, graphs)
    assert cf.same_bits(ctx.transfer_function(), lut)

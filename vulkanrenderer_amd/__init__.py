"""MI355X-native direct-volume-rendering core: Python binding of libvr_hip.so.

The product is the C-ABI library (include/vr_abi.h, vulkanrenderer_amd/csrc).  This module only
loads it with ctypes and wraps a context in a small class whose methods are named after the
reference functions they replace (Applications/VolumeRendering/Source/main.cpp).  There is no
CPU fallback: if the library is missing, importing `lib()` raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, os.environ.get("VR_LIB_NAME", "libvr_hip.so"))

DTYPE_U8, DTYPE_U16 = 1, 2
MODE_REFERENCE_ORDER, MODE_FAST, MODE_REFERENCE_SKIP = 0, 1, 2

# every symbol include/vr_abi.h declares
ABI_SYMBOLS = (
    "vr_create", "vr_destroy", "vr_last_error", "vr_stream", "vr_synchronize",
    "vr_upload_volume", "vr_upload_volume_device", "vr_load_volume_slices", "vr_volume_info",
    "vr_read_volume_normalized", "vr_histogram", "vr_set_transfer_function", "vr_get_lut",
    "vr_control_point_intensity", "vr_set_camera", "vr_set_camera_raw", "vr_get_camera",
    "vr_camera_uniforms",
    "vr_set_sampling", "vr_set_mode", "vr_set_shading", "vr_render", "vr_render_rect", "vr_render_bands", "vr_assemble_bands",
    "vr_band_rows", "vr_set_pipeline", "vr_wait_frame", "vr_march_kernel_times", "vr_tune", "vr_count_hit_rays",
    "vr_phantom_generate_device", "vr_phantom_generate_box_device",
    "vr_brick_grid", "vr_subvolume_voxels", "vr_upload_subvolume_device", "vr_render_partial",
    "vr_composite_pair", "vr_composite_ordered", "vr_decide_partials", "vr_chain_stage",
    "vr_store_chain", "vr_occupied_brick_counts", "vr_render_static", "vr_static_max_pixels", "vr_render_mapped", "vr_map_frame", "vr_unmap_frame",
)
PARTIAL_FLOATS = 5   # floats per pixel of a sort-last partial result: F0, F1, F2, Tr, S


class CtrlPt(C.Structure):
    _fields_ = [("x", C.c_float), ("intensity", C.c_float)]


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("rays_hit", C.c_uint64), ("samples_nominal", C.c_uint64),
                ("samples_evaluated", C.c_uint64), ("cells_read", C.c_uint64),
                ("bricks_touched", C.c_uint64), ("replay_rays", C.c_uint64),
                ("algorithmic_bytes", C.c_uint64), ("kernel_ms", C.c_float), ("pad_", C.c_uint32),
                ("long_rays", C.c_uint64), ("long_samples", C.c_uint64), ("replay_samples", C.c_uint64),
                ("main_bricks", C.c_uint64), ("rays_by_lane_taps", C.c_uint32 * 24), ("rays_by_wave_taps", C.c_uint32 * 24)]

    def as_dict(self):
        return {k: (list(getattr(self, k)) if k.startswith("rays_by_") else getattr(self, k)) for k, _ in self._fields_}


class PackedGraphs:
    """Four control graphs (R,G,B,A) of (x, intensity) points as the C arrays vr_set_transfer_function takes."""

    def __init__(self, graphs):
        self._keep, self.ptrs, self.counts = [], (C.POINTER(CtrlPt) * 4)(), (C.c_int * 4)()
        for c in range(4):
            g = list(graphs[c]) if c < len(graphs) and graphs[c] is not None else []
            arr = (CtrlPt * max(1, len(g)))()
            for k, (x, i) in enumerate(g):
                arr[k].x, arr[k].intensity = float(x), float(i)
            self._keep.append(arr)
            self.ptrs[c] = C.cast(arr, C.POINTER(CtrlPt))
            self.counts[c] = len(g)


class VrError(RuntimeError):
    pass


_lib = None


def lib():
    """Loads libvr_hip.so (built by __graft_entry__.build() / csrc/Makefile); no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libvr_hip.so is not built: run `make -C vulkanrenderer_amd/csrc` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    # PyTorch-ROCm bundles its own HIP runtime under the same SONAME as /opt/rocm's.  A process
    # must hold exactly one: load torch's first so that device memory, streams and RCCL handed
    # over by torch.distributed code live in the runtime this library launches its kernels on.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, fp = C.c_void_p, C.POINTER(C.c_float)
    L.vr_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.vr_destroy.argtypes = [vp]
    L.vr_last_error.argtypes = [vp]
    L.vr_last_error.restype = C.c_char_p
    L.vr_stream.argtypes = [vp]
    L.vr_stream.restype = vp
    L.vr_synchronize.argtypes = [vp]
    L.vr_upload_volume.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float]
    L.vr_upload_volume_device.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float]
    L.vr_load_volume_slices.argtypes = [vp, C.c_char_p]
    L.vr_volume_info.argtypes = [vp] + [C.POINTER(C.c_int)] * 4 + [fp]
    L.vr_read_volume_normalized.argtypes = [vp, fp]
    L.vr_histogram.argtypes = [vp, fp]
    L.vr_set_transfer_function.argtypes = [vp, C.POINTER(C.POINTER(CtrlPt)), C.POINTER(C.c_int), C.c_int]
    L.vr_get_lut.argtypes = [vp, fp]
    L.vr_control_point_intensity.argtypes = [C.c_float, C.c_float]
    L.vr_control_point_intensity.restype = C.c_float
    L.vr_set_camera.argtypes = [vp, C.c_float, C.c_float]
    L.vr_set_camera_raw.argtypes = [vp, fp, fp]
    L.vr_get_camera.argtypes = [vp, C.c_int, C.c_int, fp, fp]
    L.vr_camera_uniforms.argtypes = [C.c_float, C.c_float, C.c_int, C.c_int, fp, fp]
    L.vr_set_sampling.argtypes = [vp, C.c_int, fp]
    L.vr_set_mode.argtypes = [vp, C.c_int]
    L.vr_set_shading.argtypes = [vp, C.c_int, C.POINTER(C.c_float)]
    L.vr_render.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, C.POINTER(Stats)]
    L.vr_render_rect.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.POINTER(Stats)]
    L.vr_render_bands.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.POINTER(Stats)]
    L.vr_render_static.argtypes = [vp, C.c_int, C.c_int, vp]
    L.vr_static_max_pixels.argtypes = []
    L.vr_render_mapped.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.vr_map_frame.argtypes = [vp, C.c_int, C.POINTER(C.POINTER(C.c_uint8))]
    L.vr_unmap_frame.argtypes = [vp, C.c_int]
    L.vr_set_pipeline.argtypes = [vp, C.c_int]
    L.vr_wait_frame.argtypes = [vp, vp]
    L.vr_band_rows.argtypes = [C.c_int, C.c_int, C.c_int]
    L.vr_tune.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.vr_assemble_bands.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p]
    L.vr_march_kernel_times.argtypes = [vp, fp, C.c_int, C.POINTER(C.c_int)]
    L.vr_count_hit_rays.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint64)]
    L.vr_phantom_generate_device.argtypes = [vp, vp, C.c_int, C.c_int, C.c_uint32]
    i3, u64 = C.POINTER(C.c_int), C.c_uint64
    L.vr_phantom_generate_box_device.argtypes = [vp, vp, C.c_int, C.c_int, i3, i3, C.c_uint32]
    L.vr_brick_grid.argtypes = [C.c_int, C.c_int, C.c_int, i3]
    L.vr_subvolume_voxels.argtypes = [C.c_int, C.c_int, C.c_int, i3, i3, i3, i3]
    L.vr_upload_subvolume_device.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, i3, i3, C.c_float]
    L.vr_render_partial.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int]
    L.vr_occupied_brick_counts.argtypes = [vp, C.c_int, vp]
    L.vr_composite_pair.argtypes = [vp, C.c_int, C.c_int, u64, u64, vp, vp, C.c_int, C.c_int, vp]
    L.vr_composite_ordered.argtypes = [vp, C.c_int, C.c_int, u64, u64, vp, vp, i3, vp]
    L.vr_decide_partials.argtypes = [vp, C.c_int, C.c_int, u64, u64, vp, vp, vp, vp]
    L.vr_chain_stage.argtypes = [vp, C.c_int, C.c_int, vp, u64, vp, vp, C.c_int, i3, i3]
    L.vr_store_chain.argtypes = [vp, vp, vp, u64, u64, u64, vp]
    for name in ABI_SYMBOLS:
        fn = getattr(L, name)
        if name not in ("vr_last_error", "vr_stream", "vr_control_point_intensity"):
            fn.restype = C.c_int
    _lib = L
    return L


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def camera_uniforms(horizontal_rad, vertical_rad, width, height):
    """ViewPyramid block (20 floats) and toWorld (16 floats, column-major); host arithmetic only."""
    vp = np.empty(20, np.float32)
    m = np.empty(16, np.float32)
    rc = lib().vr_camera_uniforms(horizontal_rad, vertical_rad, width, height, _fp(vp), _fp(m))
    if rc != 0:
        raise VrError("vr_camera_uniforms failed with %d" % rc)
    return vp, m


def band_rows(height, part, parts):
    """Rows of a height-row frame that sort-first part `part` of `parts` renders."""
    n = lib().vr_band_rows(height, part, parts)
    if n < 0:
        raise VrError("vr_band_rows: bad arguments")
    return n


def band_row_indices(height, part, parts):
    """Frame row of every row of the part's dense band buffer (host arithmetic, mirrors the kernel)."""
    rows = []
    for band in range(part, (height + 15) // 16, parts):
        rows.extend(range(band * 16, min(height, band * 16 + 16)))
    return np.asarray(rows, np.int64)


def _i3(v):
    return (C.c_int * 3)(*[int(x) for x in v])


def brick_grid(nx, ny, nz):
    """Bricks per axis (x, y, z) of an nx x ny x nz volume."""
    out = (C.c_int * 3)()
    if lib().vr_brick_grid(nx, ny, nz, out) != 0:
        raise VrError("vr_brick_grid: bad arguments")
    return tuple(out)


def subvolume_voxels(nx, ny, nz, brick_lo, brick_hi):
    """Voxel box (lo, hi), each (x, y, z), that the bricks [brick_lo, brick_hi) store."""
    lo, hi = (C.c_int * 3)(), (C.c_int * 3)()
    if lib().vr_subvolume_voxels(nx, ny, nz, _i3(brick_lo), _i3(brick_hi), lo, hi) != 0:
        raise VrError("vr_subvolume_voxels: bad arguments")
    return tuple(lo), tuple(hi)


def static_max_pixels():
    return int(lib().vr_static_max_pixels())


def control_point_intensity(y, child_height):
    return float(lib().vr_control_point_intensity(y, child_height))


class VolumeRenderer:
    """One context on one GPU.  Method names follow the reference application's functions."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        rc = lib().vr_create(device, C.byref(self._h))
        if rc != 0:
            raise VrError("vr_create(%d) failed with %d" % (device, rc))

    def close(self):
        if self._h:
            lib().vr_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise VrError("%s (code %d)" % (lib().vr_last_error(self._h).decode(), rc))

    @property
    def stream(self):
        return lib().vr_stream(self._h)

    def synchronize(self):
        self._check(lib().vr_synchronize(self._h))

    # --- volume (loadVolumeData, main.cpp:112-165) -------------------------------------------
    def upload_volume(self, voxels, max_value=0.0):
        """voxels: uint8/uint16 array [nz, ny, nx] on the host."""
        voxels = np.ascontiguousarray(voxels)
        if voxels.dtype not in (np.uint8, np.uint16) or voxels.ndim != 3:
            raise TypeError("voxels must be a 3-D uint8 or uint16 array")
        nz, ny, nx = voxels.shape
        self._check(lib().vr_upload_volume(self._h, voxels.ctypes.data, voxels.dtype.itemsize, nx, ny, nz, max_value))

    def upload_volume_device(self, ptr, dtype_bytes, nx, ny, nz, max_value=0.0):
        self._check(lib().vr_upload_volume_device(self._h, ptr, dtype_bytes, nx, ny, nz, max_value))

    def load_volume_data(self, directory):
        self._check(lib().vr_load_volume_slices(self._h, os.fsencode(directory)))

    def volume_info(self):
        v = [C.c_int() for _ in range(4)]
        mx = C.c_float()
        self._check(lib().vr_volume_info(self._h, *[C.byref(x) for x in v], C.byref(mx)))
        return {"nx": v[0].value, "ny": v[1].value, "nz": v[2].value, "dtype": v[3].value, "max_value": mx.value}

    def read_volume_normalized(self):
        i = self.volume_info()
        out = np.empty((i["nz"], i["ny"], i["nx"]), np.float32)
        self._check(lib().vr_read_volume_normalized(self._h, _fp(out)))
        return out

    def prepare_histogram_volume_data(self):
        """The 100-bin intensity histogram of main.cpp:166-181."""
        out = np.empty(100, np.float32)
        self._check(lib().vr_histogram(self._h, _fp(out)))
        return out

    def phantom_generate_device(self, ptr, dtype_bytes, n, seed=0x5EED1234):
        self._check(lib().vr_phantom_generate_device(self._h, ptr, dtype_bytes, n, seed))

    # --- transfer function (updateTransferFunction, main.cpp:643-686) ------------------------
    def update_transfer_function(self, graphs, reset_state=False):
        """graphs: four sequences (R,G,B,A) of (x, intensity) control points, or a PackedGraphs (the same graphs
        handed over every frame need not be packed into C arrays every frame)."""
        packed = graphs if isinstance(graphs, PackedGraphs) else PackedGraphs(graphs)
        self._check(lib().vr_set_transfer_function(self._h, packed.ptrs, packed.counts, int(bool(reset_state))))

    def transfer_function(self):
        out = np.empty((256, 4), np.float32)
        self._check(lib().vr_get_lut(self._h, _fp(out)))
        return out

    # --- camera (setupViewPyramid / updatePixelTransformationUniforms, main.cpp:182-202, 737-799)
    def set_camera(self, horizontal_rad, vertical_rad):
        self._check(lib().vr_set_camera(self._h, horizontal_rad, vertical_rad))

    def set_camera_raw(self, location, to_world):
        loc = np.ascontiguousarray(location, np.float32)
        m = np.ascontiguousarray(to_world, np.float32)
        assert loc.size == 4 and m.size == 16
        self._check(lib().vr_set_camera_raw(self._h, _fp(loc), _fp(m)))

    def pixel_transformation_uniforms(self, width, height):
        vp = np.empty(20, np.float32)
        m = np.empty(16, np.float32)
        self._check(lib().vr_get_camera(self._h, width, height, _fp(vp), _fp(m)))
        return vp, m

    def set_sampling(self, samples_per_ray, tf_border=None):
        b = None if tf_border is None else np.ascontiguousarray(tf_border, np.float32)
        self._check(lib().vr_set_sampling(self._h, samples_per_ray, None if b is None else _fp(b)))

    def set_mode(self, mode):
        self._check(lib().vr_set_mode(self._h, mode))

    def set_shading(self, enabled, scene=None):
        """The shader's commented-out local illumination (VolumeRendering.comp:195-198); scene =
        (ambient, diffuse, specular, shininess), None = the reference's (1, 1, 1, 100) (main.cpp:90-98)."""
        sc = None if scene is None else np.ascontiguousarray(scene, np.float32)
        if sc is not None and sc.shape != (4,):
            raise ValueError("scene is (ambient, diffuse, specular, shininess)")
        self._check(lib().vr_set_shading(self._h, 1 if enabled else 0, None if sc is None else _fp(sc)))

    # --- frame (renderCommands' dispatch, main.cpp:823-828) -----------------------------------
    def render(self, width, height, stats=False, rect=None):
        x0, y0, x1, y1 = (0, 0, width, height) if rect is None else rect
        out = np.empty((y1 - y0, x1 - x0, 4), np.uint8)
        st = Stats() if stats else None
        self._check(lib().vr_render_rect(self._h, width, height, x0, y0, x1, y1, out.ctypes.data, 0,
                                         C.byref(st) if stats else None))
        return (out, st.as_dict()) if stats else out

    def tune(self, width, height, part=0, parts=1):
        """Measures and keeps the fastest per-lane work budget for the current volume, table and camera
        (on a context that holds a box of a split volume: the budget of the partial march)."""
        chosen = C.c_int()
        self._check(lib().vr_tune(self._h, width, height, part, parts, C.byref(chosen)))
        return chosen.value

    def assemble_bands(self, width, height, parts, gathered_ptr, part_stride_bytes, frame_ptr):
        """Frame from the parts' gathered band buffers (device pointers); asynchronous on the context's stream."""
        self._check(lib().vr_assemble_bands(self._h, width, height, parts, gathered_ptr, part_stride_bytes, frame_ptr))

    def render_bands(self, width, height, part, parts, ptr=None, stats=False):
        """Sort-first share of a frame: the 16-row bands b with b % parts == part, densely packed.
        With ptr (device memory) the call is asynchronous and returns only the stats."""
        rows = band_rows(height, part, parts)
        st = Stats() if stats else None
        out = None
        if ptr is None:
            out = np.empty((rows, width, 4), np.uint8)
        self._check(lib().vr_render_bands(self._h, width, height, part, parts,
                                          out.ctypes.data if ptr is None else ptr, 0 if ptr is None else 1,
                                          C.byref(st) if stats else None))
        if ptr is None:
            return (out, st.as_dict()) if stats else out
        return st.as_dict() if stats else None

    def set_pipeline(self, frames_in_flight):
        """2: fast frames into device memory overlap their finishing kernels with the next frame's march
        (alternate the output buffers and use wait_frame / synchronize before reading a frame)."""
        self._check(lib().vr_set_pipeline(self._h, frames_in_flight))

    def wait_frame(self, stream_handle):
        """Makes a stream (its hipStream_t as an integer; 0 = default stream) wait for the last frame."""
        self._check(lib().vr_wait_frame(self._h, stream_handle))

    def march_kernel_times(self, frames):
        """Device milliseconds of the march kernel of the last `frames` frames (<= 1024)."""
        ms = np.zeros(max(1, frames), np.float32)
        n = C.c_int()
        self._check(lib().vr_march_kernel_times(self._h, _fp(ms), frames, C.byref(n)))
        return ms[:n.value].copy()

    def count_hit_rays(self, width, height, rect=None):
        x0, y0, x1, y1 = (0, 0, width, height) if rect is None else rect
        n = C.c_uint64()
        self._check(lib().vr_count_hit_rays(self._h, width, height, x0, y0, x1, y1, C.byref(n)))
        return n.value

    def render_device(self, width, height, ptr, stats=False, rect=None):
        """Renders into device memory at `ptr`, asynchronously on the context's stream."""
        x0, y0, x1, y1 = (0, 0, width, height) if rect is None else rect
        st = Stats() if stats else None
        self._check(lib().vr_render_rect(self._h, width, height, x0, y0, x1, y1, ptr, 1,
                                         C.byref(st) if stats else None))
        return st.as_dict() if stats else None

    def render_static(self, width, height, ptr):
        """vr_render_static: a fast frame into device memory at `ptr` that allocates nothing (scratch from the
        library's static device array; frames up to static_max_pixels())."""
        self._check(lib().vr_render_static(self._h, width, height, ptr))

    # --- frames for a consumer on the host (the reference's copyImage hand-off, main.cpp:903-909) ---
    def render_mapped(self, width, height):
        """Renders a frame into one of the context's three slots and enqueues its copy into page-locked host
        memory beside the next frame's march; returns the slot for map_frame."""
        frame_id = C.c_int()
        self._check(lib().vr_render_mapped(self._h, width, height, C.byref(frame_id)))
        return frame_id.value

    def map_frame(self, frame_id, width, height):
        """Waits for the frame of that slot and returns a (height, width, 4) uint8 VIEW of the library's
        page-locked memory, valid until unmap_frame(frame_id)."""
        p = C.POINTER(C.c_uint8)()
        self._check(lib().vr_map_frame(self._h, frame_id, C.byref(p)))
        return np.ctypeslib.as_array(p, shape=(height, width, 4))

    def unmap_frame(self, frame_id):
        self._check(lib().vr_unmap_frame(self._h, frame_id))

    # --- sort-last: one box of the volume's brick grid per context (no reference counterpart) ---
    def phantom_generate_box_device(self, ptr, dtype_bytes, n, lo, hi, seed=0x5EED1234):
        self._check(lib().vr_phantom_generate_box_device(self._h, ptr, dtype_bytes, n, _i3(lo), _i3(hi), seed))

    def upload_subvolume_device(self, ptr, dtype_bytes, nx, ny, nz, brick_lo, brick_hi, max_value):
        self._check(lib().vr_upload_subvolume_device(self._h, ptr, dtype_bytes, nx, ny, nz, _i3(brick_lo),
                                                     _i3(brick_hi), max_value))

    def occupied_brick_counts(self, cell_bricks, counts_ptr):
        """Adds this context's occupied bricks per cube of cell_bricks^3 bricks to a device uint32 grid."""
        self._check(lib().vr_occupied_brick_counts(self._h, cell_bricks, counts_ptr))

    def render_partial(self, width, height, ptr=None):
        """Partial result of this context's box: (height*width, 5) floats.  With ptr (device memory)
        the call is asynchronous and returns None."""
        if ptr is not None:
            self._check(lib().vr_render_partial(self._h, width, height, ptr, 1))
            return None
        out = np.empty((height * width, PARTIAL_FLOATS), np.float32)
        self._check(lib().vr_render_partial(self._h, width, height, out.ctypes.data, 0))
        return out

    def composite_pair(self, width, height, first, count, mine, theirs, axis, mine_is_low, out):
        self._check(lib().vr_composite_pair(self._h, width, height, first, count, mine, theirs, axis,
                                            int(bool(mine_is_low)), out))

    def composite_ordered(self, width, height, first, count, parts, slot_of_box, grid, out):
        self._check(lib().vr_composite_ordered(self._h, width, height, first, count, parts, slot_of_box,
                                               _i3(grid), out))

    def decide_partials(self, width, height, first, count, partials, out_rgba8, undecided, undecided_count):
        self._check(lib().vr_decide_partials(self._h, width, height, first, count, partials, out_rgba8,
                                             undecided, undecided_count))

    def chain_stage(self, width, height, pixels, count, colour_in, colour_out, stage, grid, box):
        self._check(lib().vr_chain_stage(self._h, width, height, pixels, count, colour_in, colour_out, stage,
                                         _i3(grid), _i3(box)))

    def store_chain(self, pixels, colour, count, first, pixel_count, out_rgba8):
        self._check(lib().vr_store_chain(self._h, pixels, colour, count, first, pixel_count, out_rgba8))

"""ctypes binding of the CPU checker (oracle/libvr_oracle.so) — TEST INFRASTRUCTURE ONLY.

Importable from tests/, the smoke check and the benchmark's cpu_baseline leg, never from the
shipped package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libvr_oracle.so")


def build(force=False):
    """Compiles the checker with gcc (no GPU code involved)."""
    src = [os.path.join(_HERE, f) for f in ("vr_oracle.c", "vr_oracle.h", "phantom_table.inc")]
    if not force and os.path.exists(_LIB_PATH):
        if all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in src):
            return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "libvr_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class CtrlPt(C.Structure):
    _fields_ = [("x", C.c_float), ("intensity", C.c_float)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        fp = C.POINTER(C.c_float)
        L.vro_lut_update.argtypes = [fp, C.POINTER(C.POINTER(CtrlPt)), C.POINTER(C.c_int)]
        L.vro_lut_update.restype = C.c_int
        L.vro_control_point_intensity.argtypes = [C.c_float, C.c_float]
        L.vro_control_point_intensity.restype = C.c_float
        L.vro_camera.argtypes = [C.c_float, C.c_float, C.c_uint32, C.c_uint32, fp, fp]
        L.vro_camera.restype = None
        L.vro_normalize_u16.argtypes = [C.c_void_p, C.c_uint64, C.c_float, fp]
        L.vro_normalize_u16.restype = C.c_float
        L.vro_normalize_u8.argtypes = [C.c_void_p, C.c_uint64, C.c_float, fp]
        L.vro_normalize_u8.restype = C.c_float
        L.vro_load_slices.argtypes = [C.c_char_p, C.POINTER(fp), C.POINTER(C.c_uint32),
                                      C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), fp]
        L.vro_load_slices.restype = C.c_int
        L.vro_histogram.argtypes = [fp, C.c_uint64, fp]
        L.vro_histogram.restype = None
        L.vro_render_f.argtypes = [fp, C.c_uint32, C.c_uint32, C.c_uint32, fp, fp, fp, fp,
                                   C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                   C.c_void_p, C.c_void_p, C.c_int]
        L.vro_render_f.restype = None
        L.vro_render_shaded.argtypes = L.vro_render_f.argtypes + [fp]
        L.vro_render_shaded.restype = None
        L.vro_count_hits.argtypes = [fp, fp, C.c_uint32, C.c_uint32]
        L.vro_count_hits.restype = C.c_uint64
        L.vro_count_hits_rows.argtypes = [fp, fp, C.c_uint32, C.c_uint32, C.c_uint32]
        L.vro_count_hits_rows.restype = C.c_uint64
        L.vro_phantom.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_uint32]
        L.vro_phantom.restype = None
        L.vro_threads.restype = C.c_int
        i3 = C.POINTER(C.c_int)
        u32 = C.c_uint32
        L.vro_render_partial.argtypes = [fp, u32, u32, u32, fp, fp, fp, fp, u32, u32, u32, i3, i3, fp, C.c_int]
        L.vro_render_partial.restype = None
        L.vro_march_direction.argtypes = [fp, fp, u32, u32, u32, u32, u32, u32, C.c_void_p]
        L.vro_march_direction.restype = None
        L.vro_chain_stage.argtypes = [fp, u32, u32, u32, fp, fp, fp, fp, u32, u32, u32, i3, i3,
                                      C.c_void_p, u32, fp, fp, C.c_int, i3, i3]
        L.vro_chain_stage.restype = None
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def pack_points(graphs):
    """graphs: 4 sequences of (x, intensity).  Returns ctypes arrays kept alive by the caller."""
    arrays, ptrs, counts = [], (C.POINTER(CtrlPt) * 4)(), (C.c_int * 4)()
    for c in range(4):
        g = list(graphs[c]) if c < len(graphs) and graphs[c] is not None else []
        arr = (CtrlPt * max(1, len(g)))()
        for k, (x, i) in enumerate(g):
            arr[k].x, arr[k].intensity = float(x), float(i)
        arrays.append(arr)
        ptrs[c] = C.cast(arr, C.POINTER(CtrlPt))
        counts[c] = len(g)
    return arrays, ptrs, counts


def lut_update(lut, graphs):
    """Updates the 256x4 float32 table in place from four control graphs; returns the status."""
    assert lut.dtype == np.float32 and lut.size == 1024 and lut.flags.c_contiguous
    keep, ptrs, counts = pack_points(graphs)
    return lib().vro_lut_update(_fp(lut), ptrs, counts)


def control_point_intensity(y, child_h):
    return float(lib().vro_control_point_intensity(y, child_h))


def camera(h_rad, v_rad, width, height):
    vp = np.zeros(20, np.float32)
    m = np.zeros(16, np.float32)
    lib().vro_camera(h_rad, v_rad, width, height, _fp(vp), _fp(m))
    return vp, m


def normalize(vox, max_value=0.0):
    vox = np.ascontiguousarray(vox)
    out = np.empty(vox.size, np.float32)
    if vox.dtype == np.uint8:
        mx = lib().vro_normalize_u8(vox.ctypes.data, vox.size, max_value, _fp(out))
    elif vox.dtype == np.uint16:
        mx = lib().vro_normalize_u16(vox.ctypes.data, vox.size, max_value, _fp(out))
    else:
        raise TypeError(vox.dtype)
    return out.reshape(vox.shape), float(mx)


def load_slices(path):
    out = C.POINTER(C.c_float)()
    w, h, s = C.c_uint32(), C.c_uint32(), C.c_uint32()
    mx = C.c_float()
    rc = lib().vro_load_slices(path.encode(), C.byref(out), C.byref(w), C.byref(h), C.byref(s), C.byref(mx))
    if rc != 0:
        raise RuntimeError("vro_load_slices failed for %s" % path)
    n = w.value * h.value * s.value
    arr = np.ctypeslib.as_array(out, shape=(n,)).copy().reshape(s.value, h.value, w.value)
    C.CDLL(None).free(out)
    return arr, float(mx.value)


def histogram(normalized):
    a = np.ascontiguousarray(normalized, np.float32).ravel()
    h = np.zeros(100, np.float32)
    lib().vro_histogram(_fp(a), a.size, _fp(h))
    return h


def render(volume, lut, location, to_world, width, height, samples=200,
           border=(0.0, 0.0, 0.0, 0.0), rows=None, want_colour=False, threads=0, scene=None):
    """volume: float32 [nz, ny, nx] normalised; returns uint8 [height, width, 4] (and colour).
    scene = (ambient, diffuse, specular, shininess) switches on the shader's commented-out lighting."""
    volume = np.ascontiguousarray(volume, np.float32)
    nz, ny, nx = volume.shape
    lut = np.ascontiguousarray(lut, np.float32).ravel()
    loc = np.ascontiguousarray(location, np.float32).ravel()
    m = np.ascontiguousarray(to_world, np.float32).ravel()
    b = np.asarray(border, np.float32)
    out = np.zeros((height, width, 4), np.uint8)
    col = np.zeros((height, width, 3), np.float32) if want_colour else None
    r0, r1 = (0, height) if rows is None else rows
    if scene is not None:
        sc = np.asarray(scene, np.float32)
        assert sc.shape == (4,)
        lib().vro_render_shaded(_fp(volume), nx, ny, nz, _fp(lut), _fp(b), _fp(loc), _fp(m),
                                width, height, samples, r0, r1, out.ctypes.data,
                                col.ctypes.data if want_colour else None, threads, _fp(sc))
        return (out, col) if want_colour else out
    lib().vro_render_f(_fp(volume), nx, ny, nz, _fp(lut), _fp(b), _fp(loc), _fp(m),
                       width, height, samples, r0, r1, out.ctypes.data,
                       col.ctypes.data if want_colour else None, threads)
    return (out, col) if want_colour else out


def count_hits(location, to_world, width, height, rows=None):
    loc = np.ascontiguousarray(location, np.float32).ravel()
    m = np.ascontiguousarray(to_world, np.float32).ravel()
    r0, r1 = (0, height) if rows is None else rows
    return int(lib().vro_count_hits_rows(_fp(loc), _fp(m), width, r0, r1))


def phantom(n, dtype=np.uint16, seed=0x5EED1234):
    dtype = np.dtype(dtype)
    out = np.empty((n, n, n), dtype)
    lib().vro_phantom(out.ctypes.data, n, dtype.itemsize, seed)
    return out


def threads():
    return int(lib().vro_threads())


def _i3(v):
    return (C.c_int * 3)(*[int(x) for x in v])


def render_partial(volume, lut, location, to_world, width, height, samples, brick_lo, brick_hi,
                   border=(0.0, 0.0, 0.0, 0.0), threads=0):
    """Sort-last partial result of the brick box [brick_lo, brick_hi) (x, y, z): [height*width, 5]."""
    volume = np.ascontiguousarray(volume, np.float32)
    nz, ny, nx = volume.shape
    lut = np.ascontiguousarray(lut, np.float32).ravel()
    loc = np.ascontiguousarray(location, np.float32).ravel()
    m = np.ascontiguousarray(to_world, np.float32).ravel()
    b = np.asarray(border, np.float32)
    out = np.zeros((height * width, 5), np.float32)
    lib().vro_render_partial(_fp(volume), nx, ny, nz, _fp(lut), _fp(b), _fp(loc), _fp(m), width, height,
                             samples, _i3(brick_lo), _i3(brick_hi), _fp(out), threads)
    return out


def march_direction(location, to_world, width, height, samples, dims):
    """[height*width, 4] uint8: x, y, z index grows towards the eye; ray hits.  dims = (nx, ny, nz)."""
    loc = np.ascontiguousarray(location, np.float32).ravel()
    m = np.ascontiguousarray(to_world, np.float32).ravel()
    out = np.zeros((height * width, 4), np.uint8)
    lib().vro_march_direction(_fp(loc), _fp(m), width, height, samples, dims[0], dims[1], dims[2], out.ctypes.data)
    return out


def chain_stage(volume, lut, location, to_world, width, height, samples, brick_lo, brick_hi, pixels,
                colour_in, stage, grid, box, border=(0.0, 0.0, 0.0, 0.0)):
    volume = np.ascontiguousarray(volume, np.float32)
    nz, ny, nx = volume.shape
    lut = np.ascontiguousarray(lut, np.float32).ravel()
    loc = np.ascontiguousarray(location, np.float32).ravel()
    m = np.ascontiguousarray(to_world, np.float32).ravel()
    b = np.asarray(border, np.float32)
    pixels = np.ascontiguousarray(pixels, np.uint32)
    cin = np.ascontiguousarray(colour_in, np.float32)
    out = np.zeros((pixels.size, 4), np.float32)
    lib().vro_chain_stage(_fp(volume), nx, ny, nz, _fp(lut), _fp(b), _fp(loc), _fp(m), width, height, samples,
                          _i3(brick_lo), _i3(brick_hi), pixels.ctypes.data, pixels.size, _fp(cin), _fp(out),
                          stage, _i3(grid), _i3(box))
    return out

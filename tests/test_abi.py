"""The C-ABI library loads and exports every function include/vr_abi.h declares; host-only entry
points (no GPU needed) agree with the checker."""
import ctypes
import os
import re

import numpy as np

import conftest as cf
import vulkanrenderer_amd as vr


def declared_functions():
    text = open(os.path.join(cf.ROOT, "include", "vr_abi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vr_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = declared_functions()
    assert len(names) >= 20
    L = ctypes.CDLL(vr.LIB_PATH)
    for n in names:
        assert hasattr(L, n), n
    assert sorted(vr.ABI_SYMBOLS) == names


def test_camera_uniforms_bitwise_equal_checker(oracle):
    for h, v in cf.CAMERAS.values():
        for (w, hgt) in ((512, 512), (1280, 720), (3840, 2160), (33, 17)):
            vp, m = vr.camera_uniforms(h, v, w, hgt)
            ovp, om = oracle.camera(h, v, w, hgt)
            assert cf.same_bits(vp, ovp) and cf.same_bits(m, om)


def test_camera_uniforms_rejects_degenerate_extent():
    vp = np.zeros(20, np.float32)
    m = np.zeros(16, np.float32)
    fp = ctypes.POINTER(ctypes.c_float)
    assert vr.lib().vr_camera_uniforms(0.0, 0.0, 1, 720, vp.ctypes.data_as(fp), m.ctypes.data_as(fp)) == -1


def test_control_point_intensity(oracle):
    for y, hgt in ((0.0, 300.0), (299.0, 300.0), (17.0, 123.0)):
        assert vr.control_point_intensity(y, hgt) == oracle.control_point_intensity(y, hgt)


def test_binding_argument_counts_match_the_header():
    """Every prototype of the header has as many parameters as the ctypes binding declares for it."""
    text = open(os.path.join(cf.ROOT, "include", "vr_abi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    L = vr.lib()
    checked = 0
    for name, params in re.findall(r"\b(vr_[a-z_0-9]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        params = params.strip()
        count = 0 if params in ("", "void") else len(params.split(","))
        argtypes = getattr(L, name).argtypes
        assert argtypes is not None, "no argtypes for %s" % name
        assert len(argtypes) == count, (name, len(argtypes), count)
        checked += 1
    assert checked == len(vr.ABI_SYMBOLS)


def test_multi_gpu_library_exports_every_declared_symbol():
    """libvr_mg.so (include/vr_mg.h): loads, exports every declared function, binding arities match."""
    from vulkanrenderer_amd import multigpu as mg
    text = open(os.path.join(cf.ROOT, "include", "vr_mg.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = re.findall(r"\b(vrmg_[a-z_0-9]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S)
    assert sorted(n for n, _ in protos) == sorted(mg.MG_SYMBOLS)
    L = mg.lib()
    for name, params in protos:
        params = params.strip()
        count = 0 if params in ("", "void") else len(params.split(","))
        assert len(getattr(L, name).argtypes) == count, name
    assert mg.UNIQUE_ID_BYTES == 128

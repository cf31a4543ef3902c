"""ctypes binding of libvr_mg.so (include/vr_mg.h): one frame from several GPUs over RCCL, driven from C.

The reference renders on one GPU (Source/VulkanApplication.cpp:316-321); no counterpart.  This is the
same sequence as sortfirst.py / sortlast.py run over torch.distributed, behind the C ABI: a C++ host
needs nothing else.  There is no fallback: without the library `lib()` raises."""
import ctypes as C
import os

from . import VrError, _i3
from . import lib as core_lib

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvr_mg.so")
UNIQUE_ID_BYTES = 128
DIRECT_SEND, BINARY_SWAP = 0, 1
MG_SYMBOLS = ("vrmg_get_unique_id", "vrmg_create", "vrmg_create_from_comm", "vrmg_destroy", "vrmg_last_error",
              "vrmg_render_sort_first", "vrmg_wait_frame", "vrmg_synchronize", "vrmg_set_box_grid", "vrmg_render_sort_last",
              "vrmg_plan_sort_last")
MAX_ROUNDS = 16


class Round(C.Structure):
    _fields_ = [("partner", C.c_int), ("axis", C.c_int), ("mine_is_low", C.c_int), ("out_buffer", C.c_int),
                ("keep_offset", C.c_uint64), ("send_offset", C.c_uint64), ("count", C.c_uint64), ("first_pixel", C.c_uint64)]


class Plan(C.Structure):
    _fields_ = [("pixels", C.c_uint64), ("padded", C.c_uint64), ("span", C.c_uint64), ("first", C.c_uint64),
                ("box", C.c_int * 3), ("rounds", C.c_int), ("round", Round * MAX_ROUNDS)]

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libvr_mg.so is not built: run `make -C vulkanrenderer_amd/csrc`")
    core_lib()          # libvr_hip.so (and torch's HIP and RCCL, if torch is importable) first
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.vrmg_get_unique_id.argtypes = [vp]
    L.vrmg_create.argtypes = [vp, vp, C.c_int, C.c_int, C.POINTER(vp)]
    L.vrmg_create_from_comm.argtypes = [vp, vp, C.c_int, C.c_int, C.POINTER(vp)]
    L.vrmg_destroy.argtypes = [vp]
    L.vrmg_last_error.argtypes = [vp]
    L.vrmg_last_error.restype = C.c_char_p
    L.vrmg_render_sort_first.argtypes = [vp, C.c_int, C.c_int, vp]
    L.vrmg_wait_frame.argtypes = [vp, vp]
    L.vrmg_synchronize.argtypes = [vp]
    L.vrmg_plan_sort_last.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.POINTER(Plan)]
    L.vrmg_set_box_grid.argtypes = [vp, C.POINTER(C.c_int)]
    L.vrmg_render_sort_last.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, C.POINTER(C.c_uint64)]
    for name in MG_SYMBOLS:
        if name != "vrmg_last_error":
            getattr(L, name).restype = C.c_int
    _lib = L
    return L


def plan_sort_last(rank, world, grid, schedule, width, height):
    """The schedule of a sort-last frame for one rank (pure host arithmetic, usable without a GPU): a Plan."""
    plan = Plan()
    rc = lib().vrmg_plan_sort_last(rank, world, _i3(grid), schedule, width, height, C.byref(plan))
    if rc != 0:
        raise VrError("vrmg_plan_sort_last failed with %d" % rc)
    return plan


def unique_id():
    buf = C.create_string_buffer(UNIQUE_ID_BYTES)
    if lib().vrmg_get_unique_id(buf) != 0:
        raise VrError("vrmg_get_unique_id failed")
    return buf.raw


class MultiGpu:
    """One rank's end of the communicator, bound to a VolumeRenderer on this rank's GPU."""

    def __init__(self, renderer, unique_id_bytes, rank, world):
        self._h = C.c_void_p()
        self._renderer = renderer       # keeps the context alive
        rc = lib().vrmg_create(renderer._h, unique_id_bytes, rank, world, C.byref(self._h))
        if rc != 0:
            raise VrError("vrmg_create failed with %d" % rc)

    def close(self):
        if self._h:
            lib().vrmg_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise VrError("%s (code %d)" % (lib().vrmg_last_error(self._h).decode(), rc))

    def render_sort_first(self, width, height, frame_ptr):
        self._check(lib().vrmg_render_sort_first(self._h, width, height, frame_ptr))

    def wait_frame(self, stream_handle):
        """Orders a stream (its hipStream_t as an integer; 0 = default stream) after the most recent frame."""
        self._check(lib().vrmg_wait_frame(self._h, stream_handle))

    def synchronize(self):
        self._check(lib().vrmg_synchronize(self._h))

    def set_box_grid(self, grid):
        self._check(lib().vrmg_set_box_grid(self._h, _i3(grid)))

    def render_sort_last(self, width, height, schedule, frame_ptr):
        n = C.c_uint64()
        self._check(lib().vrmg_render_sort_last(self._h, width, height, schedule, frame_ptr, C.byref(n)))
        return n.value

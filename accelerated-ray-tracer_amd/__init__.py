"""accelerated-ray-tracer_amd: MI355X-native render path of the reference path tracer.

Python is only plumbing here (tests, bench, torch.distributed glue).  The
product is two C libraries built from this directory:

* ``lib/librt_mi355x.so``  -- HIP kernels + the C ABI of ``include/rt_abi.h``
* ``lib/librtw_host.so``   -- host mirror of the reference's scene classes,
  scene builders, flattener, PPM writer

This module binds both with ctypes.  There is no CPU render fallback: if the
HIP library is missing or no gfx950 device is visible, rendering raises.

The directory name carries a hyphen (it follows the reference repository's
name); import it through the ``accelerated_ray_tracer_amd`` symlink.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

PKG_DIR = os.path.dirname(os.path.realpath(__file__))
REPO_ROOT = os.path.dirname(PKG_DIR)
LIB_DIR = os.path.join(PKG_DIR, "lib")
RT_LIB_PATH = os.environ.get("RT_LIB_OVERRIDE") or os.path.join(LIB_DIR, "librt_mi355x.so")   # override: diagnostic builds only
HOST_LIB_PATH = os.environ.get("RTW_LIB_OVERRIDE") or os.path.join(LIB_DIR, "librtw_host.so")   # override: experiments only


class RtError(RuntimeError):
    pass


# ----------------------------------------------------------------------------- structs of rt_abi.h
class RtCamera(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("lower_left_corner", C.c_float * 3), ("horizontal", C.c_float * 3),
                ("vertical", C.c_float * 3), ("u", C.c_float * 3), ("v", C.c_float * 3), ("lens_radius", C.c_float),
                ("pad", C.c_float), ("time0", C.c_double), ("time1", C.c_double)]


class RtSceneDesc(C.Structure):
    _fields_ = [("nodes", C.c_void_p), ("n_nodes", C.c_int32),
                ("spheres", C.c_void_p), ("n_spheres", C.c_int32),
                ("quads", C.c_void_p), ("n_quads", C.c_int32),
                ("boxes", C.c_void_p), ("n_boxes", C.c_int32),
                ("instances", C.c_void_p), ("n_instances", C.c_int32),
                ("media", C.c_void_p), ("n_media", C.c_int32),
                ("materials", C.c_void_p), ("n_materials", C.c_int32),
                ("textures", C.c_void_p), ("n_textures", C.c_int32),
                ("images", C.c_void_p), ("image_bytes", C.c_size_t),
                ("camera", RtCamera)]


class RtFrameDesc(C.Structure):
    _fields_ = [("nx", C.c_int32), ("ny", C.c_int32), ("ns", C.c_int32), ("gamma", C.c_float),
                ("background", C.c_float * 3), ("use_gradient_bg", C.c_int32), ("seed_base", C.c_uint64),
                ("tile_rows", C.c_int32), ("tile_first", C.c_int32), ("tile_stride", C.c_int32), ("reserved", C.c_int32)]


class RtStats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("samples", C.c_uint64), ("ms_render", C.c_double), ("local_rows", C.c_int32),
                ("kernel_variant", C.c_int32), ("workgroups", C.c_int32), ("threads_per_group", C.c_int32),
                ("lds_bytes", C.c_int32), ("reserved", C.c_int32)]


NODE_DTYPE = np.dtype([("bmin", "<f4", 3), ("skip", "<i4"), ("bmax", "<f4", 3), ("prim", "<i4")])
SPHERE_DTYPE = np.dtype([("c0", "<f4", 3), ("radius", "<f4"), ("vel", "<f4", 3), ("mat", "<i4")])
MATERIAL_DTYPE = np.dtype([("kind", "<i4"), ("tex", "<i4"), ("fuzz", "<f4"), ("ior", "<f4"), ("albedo", "<f4", 3), ("pad", "<f4")])

# every symbol include/rt_abi.h declares
RT_ABI_SYMBOLS = ["rt_init", "rt_shutdown", "rt_strerror", "rt_last_hip_error", "rt_last_error_detail", "rt_scene_create",
                  "rt_scene_destroy", "rt_frame_local_rows", "rt_local_to_global_row", "rt_render", "rt_frame_finish",
                  "rt_set_option", "rt_reset_options", "rt_scene_walk_info", "rt_init_devices", "rt_multi_create", "rt_multi_render",
                  "rt_multi_destroy", "rt_multi_device_count", "rt_multi_row_owner", "rt_multi_probe_rccl", "rt_multi_debug_uninterleave",
                  "rt_progressive_state_create", "rt_progressive_state_destroy", "rt_render_window",
                  "rt_plan_walk_array", "rt_regroup_leaves"]

_rt = None
_host = None


def build(verbose: bool = False) -> None:
    """Compile both libraries and the drop-in executable in-tree (hipcc --offload-arch=gfx950)."""
    r = subprocess.run(["make", "-C", PKG_DIR, "-j8", "all"], capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout[-4000:])
        print(r.stderr[-4000:])
    if r.returncode != 0:
        raise RtError("building the HIP render library failed")


def host_lib():
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise RtError(f"{HOST_LIB_PATH} not built (run __graft_entry__.build())")
        L = C.CDLL(HOST_LIB_PATH)
        L.rtw_last_error.restype = C.c_char_p
        L.rtw_scene_name.restype = C.c_char_p
        L.rtw_scene_name.argtypes = [C.c_int]
        L.rtw_scene_build.restype = C.c_void_p
        L.rtw_scene_build.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
        L.rtw_scene_free.argtypes = [C.c_void_p]
        L.rtw_scene_desc.restype = C.POINTER(RtSceneDesc)
        L.rtw_scene_desc.argtypes = [C.c_void_p]
        L.rtw_scene_defaults.restype = C.c_float
        L.rtw_scene_defaults.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_int)]
        L.rtw_scene_leaf_order.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.rtw_write_ppm.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.rtw_load_ppm.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        _host = L
    return _host


def rt_lib():
    """The HIP render library.  Raises when it has not been built -- there is nothing to fall back to."""
    global _rt
    if _rt is None:
        if not os.path.exists(RT_LIB_PATH):
            raise RtError(f"{RT_LIB_PATH} not built (run __graft_entry__.build()); the render path has no CPU fallback")
        L = C.CDLL(RT_LIB_PATH)
        L.rt_init.argtypes = [C.c_int]
        L.rt_strerror.restype = C.c_char_p
        L.rt_strerror.argtypes = [C.c_int]
        L.rt_last_error_detail.restype = C.c_char_p
        L.rt_scene_create.argtypes = [C.POINTER(RtSceneDesc), C.POINTER(C.c_void_p)]
        L.rt_scene_destroy.argtypes = [C.c_void_p]
        L.rt_frame_local_rows.argtypes = [C.POINTER(RtFrameDesc)]
        L.rt_local_to_global_row.argtypes = [C.POINTER(RtFrameDesc), C.c_int32]
        L.rt_render.argtypes = [C.c_void_p, C.POINTER(RtFrameDesc), C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(RtStats)]
        L.rt_frame_finish.argtypes = [C.c_void_p, C.POINTER(RtStats)]
        L.rt_set_option.argtypes = [C.c_char_p, C.c_int]
        L.rt_init_devices.argtypes = [C.c_int]
        L.rt_multi_create.argtypes = [C.POINTER(RtSceneDesc), C.c_int, C.POINTER(C.c_void_p)]
        L.rt_multi_render.argtypes = [C.c_void_p, C.POINTER(RtFrameDesc), C.c_void_p, C.c_int, C.c_int, C.POINTER(RtStats)]
        L.rt_multi_destroy.argtypes = [C.c_void_p]
        L.rt_multi_device_count.argtypes = [C.c_void_p]
        L.rt_multi_row_owner.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.rt_progressive_state_create.argtypes = [C.c_void_p, C.POINTER(RtFrameDesc), C.POINTER(C.c_void_p)]
        L.rt_progressive_state_destroy.argtypes = [C.c_void_p, C.c_void_p]
        L.rt_render_window.argtypes = [C.c_void_p, C.POINTER(RtFrameDesc), C.c_void_p, C.c_int, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int, C.POINTER(RtStats)]
        L.rt_multi_probe_rccl.argtypes = [C.c_char_p]
        L.rt_multi_debug_uninterleave.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
        L.rt_plan_walk_array.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_double, C.c_void_p, C.c_int32, C.POINTER(C.c_int32),
                                         C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.rt_regroup_leaves.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
        L.rt_regroup_leaves.restype = C.c_int
        L.rt_scene_walk_info.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        _rt = L
    return _rt


def _check(st: int, what: str) -> None:
    if st != 0:
        L = rt_lib()
        raise RtError(f"{what}: {L.rt_strerror(st).decode()} -- {L.rt_last_error_detail().decode()}")


def load_ppm(path: str):
    """Texture pixels from a PPM file -> (uint8 array h*w*3, w, h)."""
    L = host_lib()
    w, h = C.c_int(0), C.c_int(0)
    n = L.rtw_load_ppm(path.encode(), None, 0, C.byref(w), C.byref(h))
    if n < 0:
        raise RtError(f"cannot read PPM {path}")
    buf = np.zeros(n, np.uint8)
    L.rtw_load_ppm(path.encode(), buf.ctypes.data, n, C.byref(w), C.byref(h))
    return buf, w.value, h.value


SCENE_TEXTURES = {"earth": "earthmap.ppm", "final": "earthmap.ppm", "simple_light": "poolball.ppm", "original": "8ball.ppm"}


def default_texture(scene: str = "final"):
    """The image a reference scene loads (main.cu:816,1010,1186,1254), decoded once to assets/*.ppm; (None,0,0) otherwise."""
    name = SCENE_TEXTURES.get(scene)
    if name is None:
        return None, 0, 0
    p = os.path.join(REPO_ROOT, "assets", name)
    return load_ppm(p) if os.path.exists(p) else (None, 0, 0)


class HostScene:
    """A reference scene built and flattened on the host (no GPU involved)."""

    def __init__(self, name: str, nx: int = 0, ny: int = 0, image=None, iw: int = 0, ih: int = 0):
        L = host_lib()
        self._image = None if image is None else np.ascontiguousarray(image, np.uint8)
        ptr = None if self._image is None else self._image.ctypes.data
        self._h = L.rtw_scene_build(name.encode(), nx, ny, ptr, iw, ih)
        if not self._h:
            raise RtError(f"scene '{name}': {L.rtw_last_error().decode()}")
        self.name = name
        self.desc = L.rtw_scene_desc(self._h).contents
        out4 = (C.c_int * 4)()
        bg = (C.c_float * 3)()
        dbl = C.c_int(0)
        self.gamma = float(L.rtw_scene_defaults(self._h, out4, bg, C.byref(dbl)))
        self.nx, self.ny, self.ns, self.use_gradient_bg = (int(x) for x in out4)
        self.background = [float(x) for x in bg]
        self.ppm_double_scale = bool(dbl.value)

    def nodes(self) -> np.ndarray:
        n = self.desc.n_nodes
        buf = (C.c_char * (n * NODE_DTYPE.itemsize)).from_address(self.desc.nodes)
        return np.frombuffer(buf, NODE_DTYPE, n).copy()

    def spheres(self) -> np.ndarray:
        n = self.desc.n_spheres
        buf = (C.c_char * (n * SPHERE_DTYPE.itemsize)).from_address(self.desc.spheres)
        return np.frombuffer(buf, SPHERE_DTYPE, n).copy()

    def materials(self) -> np.ndarray:
        n = self.desc.n_materials
        buf = (C.c_char * (n * MATERIAL_DTYPE.itemsize)).from_address(self.desc.materials)
        return np.frombuffer(buf, MATERIAL_DTYPE, n).copy()

    def leaf_order(self) -> np.ndarray:
        L = host_lib()
        n = self.desc.n_nodes
        out = np.zeros(n, np.int32)
        L.rtw_scene_leaf_order(self._h, out.ctypes.data, n)
        return out

    def frame(self, nx=None, ny=None, ns=None, gamma=None, seed_base=1984, tile_rows=None, tile_first=0, tile_stride=1) -> RtFrameDesc:
        f = RtFrameDesc()
        f.nx = self.nx if nx is None else nx
        f.ny = self.ny if ny is None else ny
        f.ns = self.ns if ns is None else ns
        f.gamma = self.gamma if gamma is None else gamma
        f.background[:] = self.background
        f.use_gradient_bg = self.use_gradient_bg
        f.seed_base = seed_base
        f.tile_rows = f.ny if tile_rows is None else tile_rows
        f.tile_first = tile_first
        f.tile_stride = tile_stride
        return f

    def close(self):
        if getattr(self, "_h", None):
            host_lib().rtw_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_initialised_device = None


def init(device: int = 0) -> None:
    global _initialised_device
    _check(rt_lib().rt_init(device), "rt_init")
    _initialised_device = device


def set_option(key: str, value: int) -> None:
    _check(rt_lib().rt_set_option(key.encode(), int(value)), f"rt_set_option({key})")


def reset_options() -> None:
    """Every scheduling knob back to the shipped default."""
    _check(rt_lib().rt_reset_options(), "rt_reset_options")


class DeviceScene:
    """rt_scene*: the flattened scene resident in HBM."""

    def __init__(self, host_scene: HostScene):
        if _initialised_device is None:
            init(0)
        self.host = host_scene
        self._p = C.c_void_p()
        _check(rt_lib().rt_scene_create(C.byref(host_scene.desc), C.byref(self._p)), "rt_scene_create")

    def render(self, frame: RtFrameDesc, out=None, stream: int = 0, blocking: bool = True):
        """Render into `out`: a float32 numpy array (host) or an integer device pointer.  Returns (array|None, stats)."""
        L = rt_lib()
        rows = L.rt_frame_local_rows(C.byref(frame))
        if rows < 0:
            raise RtError("bad row partition")
        stats = RtStats()
        if out is None:
            out = np.empty((rows, frame.nx, 3), np.float32)
        if isinstance(out, np.ndarray):
            assert out.dtype == np.float32 and out.size == rows * frame.nx * 3 and out.flags["C_CONTIGUOUS"]
            _check(L.rt_render(self._p, C.byref(frame), out.ctypes.data, 0, stream, 1, C.byref(stats)), "rt_render")
            return out, stats
        _check(L.rt_render(self._p, C.byref(frame), C.c_void_p(int(out)), 1, stream, 1 if blocking else 0, C.byref(stats)), "rt_render")
        return None, stats

    def progressive(self, frame: RtFrameDesc) -> "ProgressiveFrame":
        """Progressive accumulation of `frame` (rt_render_window): windows of samples, a displayable frame after each."""
        return ProgressiveFrame(self, frame)

    def walk_info(self) -> dict:
        """Node counts of the reference tree / the walk array and expected box tests per ray on the calibration frame."""
        a, b, x, y = C.c_int32(0), C.c_int32(0), C.c_double(0), C.c_double(0)
        _check(rt_lib().rt_scene_walk_info(self._p, C.byref(a), C.byref(b), C.byref(x), C.byref(y)), "rt_scene_walk_info")
        return {"nodes_reference": a.value, "nodes_walked": b.value, "tests_before": x.value, "tests_after": y.value}

    def finish(self) -> RtStats:
        stats = RtStats()
        _check(rt_lib().rt_frame_finish(self._p, C.byref(stats)), "rt_frame_finish")
        return stats

    def close(self):
        if self._p:
            rt_lib().rt_scene_destroy(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ProgressiveFrame:
    """rt_progressive_state_* + rt_render_window: the per-pixel XORWOW state and colour sum carried between windows."""

    def __init__(self, scene: DeviceScene, frame: RtFrameDesc):
        self.scene, self.frame = scene, frame
        self._p = C.c_void_p()
        _check(rt_lib().rt_progressive_state_create(scene._p, C.byref(frame), C.byref(self._p)), "rt_progressive_state_create")
        self.rows = rt_lib().rt_frame_local_rows(C.byref(frame))

    def render(self, sample_begin: int, sample_end: int):
        """Samples [sample_begin, sample_end) of every pixel; returns (the frame averaged over sample_end samples, stats)."""
        out = np.empty((self.rows, self.frame.nx, 3), np.float32)
        stats = RtStats()
        _check(rt_lib().rt_render_window(self.scene._p, C.byref(self.frame), out.ctypes.data, 0, self._p, sample_begin, sample_end, None, 1, C.byref(stats)), "rt_render_window")
        return out, stats

    def close(self):
        if self._p:
            rt_lib().rt_progressive_state_destroy(self.scene._p, self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiScene:
    """rt_multi*: one replica of the scene per GPU, driven from this thread (rt_multi_* of include/rt_abi.h)."""

    def __init__(self, host_scene: HostScene, n_gpus: int):
        global _initialised_device
        _check(rt_lib().rt_init_devices(n_gpus), "rt_init_devices")
        _initialised_device = 0
        self.host = host_scene
        self._p = C.c_void_p()
        _check(rt_lib().rt_multi_create(C.byref(host_scene.desc), n_gpus, C.byref(self._p)), "rt_multi_create")

    def render(self, frame: RtFrameDesc, tile_rows: int = 4):
        """The whole frame as float32[ny][nx][3] in host memory, and the summed statistics."""
        out = np.empty((frame.ny, frame.nx, 3), np.float32)
        stats = RtStats()
        _check(rt_lib().rt_multi_render(self._p, C.byref(frame), out.ctypes.data, 0, tile_rows, C.byref(stats)), "rt_multi_render")
        return out, stats

    def close(self):
        if self._p:
            rt_lib().rt_multi_destroy(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def plan_walk_array(nodes: np.ndarray, passes=None, root_visits: float = 0.0):
    """The walk array rt_scene_create would derive from `nodes` (NODE_DTYPE) for the given per-node pass counts (None = by
    surface area).  Host only.  Returns (walk nodes, tests per ray before, after)."""
    nodes = np.ascontiguousarray(nodes, NODE_DTYPE)
    n = len(nodes)
    out = np.zeros(n, NODE_DTYPE)
    n_out, tb, ta = C.c_int32(0), C.c_double(0), C.c_double(0)
    p = None if passes is None else np.ascontiguousarray(passes, np.float64)
    _check(rt_lib().rt_plan_walk_array(nodes.ctypes.data, n, None if p is None else p.ctypes.data, float(root_visits), out.ctypes.data, n,
                                       C.byref(n_out), C.byref(tb), C.byref(ta)), "rt_plan_walk_array")
    return out[: n_out.value].copy(), tb.value, ta.value


def regroup_leaves(nodes: np.ndarray, method: int = 0) -> np.ndarray:
    """The leaves of `nodes` (NODE_DTYPE) under another binary tree over the same leaf order (method 0 = top-down by
    surface-area cost, 1 = bottom-up).  Host only."""
    nodes = np.ascontiguousarray(nodes, NODE_DTYPE)
    out = np.zeros(2 * len(nodes), NODE_DTYPE)
    n_out = C.c_int32(0)
    _check(rt_lib().rt_regroup_leaves(nodes.ctypes.data, len(nodes), int(method), out.ctypes.data, len(out), C.byref(n_out)), "rt_regroup_leaves")
    return out[: n_out.value].copy()


def row_owner(global_row: int, tile_rows: int, n_gpus: int):
    d, l = C.c_int32(0), C.c_int32(0)
    _check(rt_lib().rt_multi_row_owner(global_row, tile_rows, n_gpus, C.byref(d), C.byref(l)), "rt_multi_row_owner")
    return d.value, l.value


def local_rows_to_global(frame: RtFrameDesc) -> np.ndarray:
    L = rt_lib()
    rows = L.rt_frame_local_rows(C.byref(frame))
    return np.array([L.rt_local_to_global_row(C.byref(frame), k) for k in range(rows)], np.int64)


def write_ppm(path: str, fb: np.ndarray, double_scale: bool = False, binary: bool = False) -> None:
    """ASCII P3 as the reference prints it (main.cu:715-727, unclamped), or binary P6 (clamped to 0..255)."""
    fb = np.ascontiguousarray(fb, np.float32)
    ny, nx = fb.shape[0], fb.shape[1]
    if host_lib().rtw_write_ppm(path.encode(), fb.ctypes.data, nx, ny, (1 if double_scale else 0) | (2 if binary else 0)) != 0:
        raise RtError(f"cannot write {path}")

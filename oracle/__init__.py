"""CPU oracle binding.  TEST INFRASTRUCTURE ONLY (see rt_oracle.cpp header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this;
nothing in accelerated-ray-tracer_amd/ does.  The reference has no golden vectors and
cannot be built in this image; the oracle is pinned at 8-bit level against the output images
the reference holds (tests/test_reference_images.py) and by the counters SURVEY.md recorded
from the reference's own code (tests/golden/survey_pins.json).  Below 8 bits (fp32 bit-level
vs a CUDA build) parity is unpinned: no such output of the reference exists.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.realpath(__file__))
LIB_PATH = os.path.join(_DIR, "librt_oracle.so")
_lib = None

COUNTER_NAMES = ["rays", "box_tests", "sphere_tests", "quad_tests", "medium_calls", "box6_calls", "inst_calls", "samples"]
CENSUS_NAMES = ["list", "spheres", "moving_spheres", "quads", "boxes", "instances", "media", "lambertian", "metal",
                "dielectric", "light", "depth"]


def build() -> None:
    r = subprocess.run(["make", "-C", _DIR, "librt_oracle.so"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + r.stdout + r.stderr)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.orc_scene_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
        L.orc_scene_defaults.restype = C.c_float
        L.orc_scene_defaults.argtypes = [C.c_int, C.c_void_p]
        L.orc_render.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int,
                                 C.c_ulonglong, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_xorwow.argtypes = [C.c_ulonglong, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_dump_nodes.argtypes = [C.c_int, C.c_void_p, C.c_int]
        L.orc_scene_census.argtypes = [C.c_int, C.c_void_p]
        L.orc_camera.argtypes = [C.c_int, C.c_void_p]
        L.orc_perlin_noise.restype = C.c_float
        L.orc_perlin_noise.argtypes = [C.c_float] * 3
        L.orc_perlin_turb.restype = C.c_float
        L.orc_perlin_turb.argtypes = [C.c_float, C.c_float, C.c_float, C.c_int]
        _lib = L
    return _lib


def xorwow(seed: int, n: int):
    state = np.zeros(6, np.uint32)
    uni = np.zeros(n, np.float32)
    raw = np.zeros(n, np.uint32)
    lib().orc_xorwow(seed, n, state.ctypes.data, uni.ctypes.data, raw.ctypes.data)
    return state, uni, raw


class OracleScene:
    def __init__(self, name: str, nx: int, ny: int, image=None, iw: int = 0, ih: int = 0):
        self._img = None if image is None else np.ascontiguousarray(image, np.uint8)
        ptr = None if self._img is None else self._img.ctypes.data
        self.h = lib().orc_scene_create(name.encode(), nx, ny, ptr, iw, ih)
        if self.h < 0:
            raise ValueError(f"oracle has no scene '{name}'")
        self.name, self.nx, self.ny = name, nx, ny
        d = np.zeros(7, np.float32)
        self.gamma = float(lib().orc_scene_defaults(self.h, d.ctypes.data))
        self.def_nx, self.def_ny, self.def_ns, self.gradient = int(d[0]), int(d[1]), int(d[2]), int(d[3])
        self.background = d[4:7].copy()

    def render(self, ns: int, gamma=None, seed_base: int = 1984, row0: int = 0, row1=None, threads: int = 0, counters: bool = True):
        """Full-frame layout float32[ny][nx][3] (row 0 = bottom); rows outside [row0,row1) stay zero."""
        row1 = self.ny if row1 is None else row1
        threads = threads or min(os.cpu_count() or 1, 16)
        fb = np.zeros((self.ny, self.nx, 3), np.float32)
        cnt = np.zeros(8, np.uint64)
        bg = np.ascontiguousarray(self.background, np.float32)
        lib().orc_render(self.h, fb.ctypes.data, self.nx, self.ny, ns, self.gamma if gamma is None else gamma, bg.ctypes.data,
                         self.gradient, seed_base, row0, row1, cnt.ctypes.data if counters else None, threads)
        return fb, dict(zip(COUNTER_NAMES, (int(x) for x in cnt)))

    def nodes(self) -> np.ndarray:
        n = lib().orc_dump_nodes(self.h, None, 0)
        out = np.zeros((n, 8), np.float32)
        lib().orc_dump_nodes(self.h, out.ctypes.data, n)
        return out

    def node_passes(self, ns: int, threads: int = 0) -> tuple:
        """Diagnostics: renders the frame at `ns` spp counting, per BVH node (depth-first pre-order, as nodes()), the box tests
        that passed.  Returns (passes per node, rays, box tests)."""
        L = lib()
        L.orc_node_stats_enable.argtypes = [C.c_int, C.c_int]
        L.orc_node_passes.argtypes = [C.c_int, C.c_void_p, C.c_int]
        L.orc_node_stats_enable(self.h, 1)
        try:
            _, cnt = self.render(ns, threads=threads)
            n = len(self.nodes())
            p = np.zeros(n, np.uint64)
            L.orc_node_passes(self.h, p.ctypes.data, n)
        finally:
            L.orc_node_stats_enable(self.h, 0)
        return p.astype(np.float64), cnt["rays"], cnt["box_tests"]

    def census(self) -> dict:
        c = np.zeros(12, np.int32)
        lib().orc_scene_census(self.h, c.ctypes.data)
        return dict(zip(CENSUS_NAMES, (int(x) for x in c)))

    def camera(self) -> np.ndarray:
        c = np.zeros(21, np.float32)
        lib().orc_camera(self.h, c.ctypes.data)
        return c

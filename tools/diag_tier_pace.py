#!/usr/bin/env python3
"""Cycles per ray in the tier kernel (diagnostic build: RT_LIB_OVERRIDE=.../librt_mi355x_diag.so).
The first wave of every tier workgroup times its trace_wave calls and its resolve + shade with the shader clock.
SCENE / NX / NY / NS / STRIDE select the frame (default: the headline frame, whole).
Usage: diag_tier_pace.py [key=value ...]        e.g. tier_auto=0 tier1_pixels=4096 tier1_factor_x10=30"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import accelerated_ray_tracer_amd as art
art.init(0)
for kv in sys.argv[1:]:
    k, v = kv.split("="); art.set_option(k, int(v))
scene, nx, ny, ns, stride = os.environ.get("SCENE", "random_scene"), int(os.environ.get("NX", "1200")), int(os.environ.get("NY", "800")), int(os.environ.get("NS", "500")), int(os.environ.get("STRIDE", "1"))
img, iw, ih = art.default_texture(scene)
hs = art.HostScene(scene, nx, ny, img, iw, ih); ds = art.DeviceScene(hs)
fb, st = ds.render(hs.frame(nx=nx, ny=ny, ns=ns, tile_rows=4 if stride > 1 else ny, tile_first=0, tile_stride=stride))
c = np.zeros(16, np.uint64); L = art.rt_lib(); L.rt_debug_counters.argtypes = [C.c_void_p, C.c_void_p]; L.rt_debug_counters(ds._p, c.ctypes.data)
c = [int(x) for x in c]
print(scene, f"{nx}x{ny}@{ns} 1/{stride}", sys.argv[1:], "frame %.1f ms (diag build)" % st.ms_render, "tier rays timed", c[15], "traversal cycles per ray %.0f" % (c[14] / max(c[15], 1)))

if hasattr(L, "rt_debug_stage_cycles"):
    t = np.zeros(10, np.uint64); L.rt_debug_stage_cycles.argtypes = [C.c_void_p, C.c_void_p]
    if L.rt_debug_stage_cycles(ds._p, t.ctypes.data) == 0:
        print("   tier loops, first wave of each tier workgroup: resolve + shade %.0f cycles per ray, whole loop %.0f cycles per ray (traversal %.0f)" % (int(t[8]) / max(c[15], 1), int(t[9]) / max(c[15], 1), c[14] / max(c[15], 1)))

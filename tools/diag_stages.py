#!/usr/bin/env python3
"""Per-stage execution counts of the staged kernel (diagnostic build lib/librt_mi355x_diag.so, -DRT_DIAG).
Run with RT_LIB_OVERRIDE=accelerated-ray-tracer_amd/lib/librt_mi355x_diag.so.  Timings of this build are not quoted."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import accelerated_ray_tracer_amd as art
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 20
scene, nx, ny = os.environ.get("SCENE", "random_scene"), int(os.environ.get("NX", "1200")), int(os.environ.get("NY", "800"))
opts = dict(kv.split("=") for kv in sys.argv[2:])
art.init(0)
for k, v in opts.items(): art.set_option(k, int(v))
art.set_option("kernel", 3)
img, iw, ih = art.default_texture(scene)
hs = art.HostScene(scene, nx, ny, img, iw, ih)
ds = art.DeviceScene(hs)
stride, first = int(os.environ.get("STRIDE", "1")), int(os.environ.get("FIRST", "0"))   # one rank's share of a STRIDE-GPU run
fb, st = ds.render(hs.frame(nx=nx, ny=ny, ns=ns, tile_rows=4 if stride > 1 else ny, tile_first=first, tile_stride=stride))
c = np.zeros(16, np.uint64)
L = art.rt_lib(); L.rt_debug_counters.argtypes = [C.c_void_p, C.c_void_p]
L.rt_debug_counters(ds._p, c.ctypes.data)
c = [int(x) for x in c]
rays = st.rays
names = ["trips", "box_steps(wave)", "box_lane_steps", "leaf_passes", "leaf_lanes", "stageC_runs", "stageC_lanes", "stageD_runs", "stageD_lanes",
         "stageE_runs", "stageE_lanes", "stageC_hit_lanes", "stageF_runs", "stageF_lanes"]
print(f"scene {scene} {nx}x{ny}@{ns} variant {st.kernel_variant} wgs {st.workgroups} x {st.threads_per_group}")
print(f"rays {rays}  wave-rays {rays/64:.0f}  ms {st.ms_render:.2f}  opts {opts}")
for n, v in zip(names, c): print(f"  {n:18s} {v:14d}   per ray {v/rays:8.4f}   per wave-ray(64) {v/(rays/64):8.3f}")
print(f"  box lanes/step {c[2]/max(c[1],1):.1f}  leaf lanes/pass {c[4]/max(c[3],1):.1f}  C lanes/run {c[6]/max(c[5],1):.1f} (hits {c[11]/max(c[5],1):.1f})  D lanes/run {c[8]/max(c[7],1):.1f}  E lanes/run {c[10]/max(c[9],1):.1f}  F lanes/run {c[13]/max(c[12],1):.1f}")

# cycles per part of the loop, summed over waves (diagnostic build only)
if hasattr(L, "rt_debug_stage_cycles"):
    t = np.zeros(10, np.uint64); L.rt_debug_stage_cycles.argtypes = [C.c_void_p, C.c_void_p]
    if L.rt_debug_stage_cycles(ds._p, t.ctypes.data) == 0:
        t = [int(x) for x in t][:8]; tot = max(sum(t), 1)
        labels = ["box steps", "object tests", "stage C", "stage D", "stage E", "-", "stage gating", "stage F + loop"]
        print("  wave cycles by part of the loop: " + "  ".join(f"{l} {100.0 * v / tot:.1f} %" for l, v in zip(labels, t) if l != "-"))
        print(f"  cycles per wave-ray(64): {tot / (rays / 64):.0f}")

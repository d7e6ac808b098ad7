#!/usr/bin/env python3
"""Wall time of rt_scene_create (upload + calibration pass + walk-array plan) per BASELINE scene, and the walk it produced."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import accelerated_ray_tracer_amd as art
art.init(0)
for scene, nx, ny in (("random_scene", 1200, 800), ("cornell", 600, 600), ("cornell_smoke", 600, 600), ("final", 800, 800)):
    img, iw, ih = art.default_texture(scene)
    hs = art.HostScene(scene, nx, ny, img, iw, ih)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); ds = art.DeviceScene(hs); ts.append((time.perf_counter() - t0) * 1e3)
        info = ds.walk_info(); ds.close()
    print(f"{scene:14s} rt_scene_create {min(ts):8.2f} ms (first {ts[0]:.1f})  walk {info}", flush=True)

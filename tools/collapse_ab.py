#!/usr/bin/env python3
"""Frame time by walk-array mode (option bvh_collapse, read at scene creation): one scene object per mode, interleaved renders,
identical frames asserted.  Usage: [SCENE= NX= NY= NS=] collapse_ab.py [modes...]   (default modes: 2 3)"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import accelerated_ray_tracer_amd as art
scene, nx, ny, ns = os.environ.get("SCENE", "random_scene"), int(os.environ.get("NX", "1200")), int(os.environ.get("NY", "800")), int(os.environ.get("NS", "500"))
modes = [int(x) for x in sys.argv[1:]] or [2, 3]
art.init(0)
img, iw, ih = art.default_texture(scene)
hs = art.HostScene(scene, nx, ny, img, iw, ih)
scenes = {}
for m in modes:
    art.reset_options(); art.set_option("bvh_collapse", m)
    scenes[m] = art.DeviceScene(hs)
art.reset_options()
buf = torch.zeros((ny, nx, 3), dtype=torch.float32, device="cuda")
times = {m: [] for m in modes}; digest = {}
for rnd in range(4):
    for m in modes:
        buf.zero_()
        _, st = scenes[m].render(hs.frame(nx=nx, ny=ny, ns=ns), out=buf.data_ptr(), blocking=True)
        times[m].append(st.ms_render)
        if rnd == 0: digest[m] = hashlib.sha1(buf.cpu().numpy().tobytes()).hexdigest()[:12]
for m in modes:
    info = scenes[m].walk_info()
    print(f"{scene} {nx}x{ny}@{ns} bvh_collapse={m}: min {min(times[m]):8.3f} ms  med {float(np.median(times[m])):8.3f}  walk {info['nodes_walked']} nodes, {info['tests_after']:.2f} box tests/ray  frame {'same' if digest[m] == digest[modes[0]] else 'DIFFERENT'}", flush=True)

#!/usr/bin/env python3
"""A/B sweep of kernel variants / knobs in ONE process (interleaved rounds), checking that every configuration
produces the identical frame.  Usage: sweep.py [--scene S --nx --ny --ns --rounds R] cfg1 cfg2 ...
where cfg is comma-separated key=value rt_set_option pairs, e.g.  kernel=2,threads=256,wg_per_cu=3"""
import argparse, os, sys, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import accelerated_ray_tracer_amd as art

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="random_scene"); ap.add_argument("--nx", type=int, default=1200); ap.add_argument("--ny", type=int, default=800)
ap.add_argument("--ns", type=int, default=50); ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("cfgs", nargs="+")
a = ap.parse_args()
art.init(0)
img, iw, ih = art.default_texture(a.scene)
hs = art.HostScene(a.scene, a.nx, a.ny, img, iw, ih)
ds = art.DeviceScene(hs)
frame = hs.frame(nx=a.nx, ny=a.ny, ns=a.ns)
buf = torch.zeros((a.ny, a.nx, 3), dtype=torch.float32, device="cuda")
times = {c: [] for c in a.cfgs}; digest = {}; rays = {}; variant = {}; tail = {}
for rnd in range(a.rounds):
    for c in a.cfgs:
        art.reset_options()
        for kv in c.split(","):
            if kv: k, v = kv.split("="); art.set_option(k, int(v))
        buf.zero_()
        _, st = ds.render(frame, out=buf.data_ptr(), blocking=True)
        times[c].append(st.ms_render); rays[c] = st.rays; variant[c] = (st.kernel_variant, st.workgroups, st.threads_per_group, st.lds_bytes, st.reserved)
        if rnd == 0:
            digest[c] = hashlib.sha1(buf.cpu().numpy().tobytes()).hexdigest()[:12]
            L = art.rt_lib()
            if hasattr(L, "rt_debug_handoff"):     # pixels handed to the tail launches, samples they had to go
                import ctypes as C
                h = np.zeros(2, np.uint64); L.rt_debug_handoff.argtypes = [C.c_void_p, C.c_void_p]
                if L.rt_debug_handoff(ds._p, h.ctypes.data) == 0: tail[c] = ("tail", int(h[0]), int(h[1]))
ref = digest[a.cfgs[0]]
for c in a.cfgs:
    t = times[c]
    print(f"{c:60s} min {min(t):9.3f} ms  med {float(np.median(t)):9.3f} ms  {rays[c]/min(t)/1e3:9.1f} Mrays/s  frame {'same' if digest[c]==ref else 'DIFFERENT '+digest[c]}  {variant[c]} {tail.get(c, '')}", flush=True)

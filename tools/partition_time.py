#!/usr/bin/env python3
"""Times one rank's share of the headline frame on one GPU: rows of tiles tile_first, tile_first + stride, ... (what rank
tile_first of a `stride`-GPU run renders).  With no gather this is the rank-local time of a multi-GPU run, which is
bound by the dearest pixels' sequential chains.  Usage: partition_time.py [stride ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import accelerated_ray_tracer_amd as art
art.init(0)
for kv in filter(None, os.environ.get("RT_OPTS", "").split(",")):   # A/B knobs: RT_OPTS=key=value,key=value
    k, v = kv.split("="); art.set_option(k, int(v))
# SCENE / NX / NY / NS select another BASELINE configuration (defaults: the headline frame)
scene, nx, ny, ns = os.environ.get("SCENE", "random_scene"), int(os.environ.get("NX", "1200")), int(os.environ.get("NY", "800")), int(os.environ.get("NS", "500"))
img, iw, ih = art.default_texture(scene)
hs = art.HostScene(scene, nx, ny, img, iw, ih)
ds = art.DeviceScene(hs)
print(f"# {scene} {nx}x{ny} @ {ns} spp, 4-row tiles dealt round-robin; rank-local render time on one MI355X (no gather)", flush=True)
t1 = None
for stride in [int(x) for x in (sys.argv[1:] or ["1", "2", "4", "8"])]:
    worst = 0.0
    for first in range(stride):
        f = hs.frame(nx=nx, ny=ny, ns=ns, tile_rows=4 if stride > 1 else ny, tile_first=first, tile_stride=stride)
        rows = art.rt_lib().rt_frame_local_rows(f)
        buf = torch.zeros((rows, nx, 3), dtype=torch.float32, device="cuda")
        ts = []
        for _ in range(3):
            _, st = ds.render(f, out=buf.data_ptr(), blocking=True)
            ts.append(st.ms_render)
        worst = max(worst, min(ts))
        print(f"stride {stride} rank {first}: {min(ts):8.3f} ms  {st.rays/1e6:8.1f} Mrays  heavy {st.reserved}", flush=True)
    if t1 is None: t1 = worst
    print(f"== N={stride}: slowest rank {worst:.3f} ms   T1/(N*T_N) = {t1 / (stride * worst):.3f}", flush=True)

#!/usr/bin/env python3
"""Renders FRAMES (default 3) frames of one configuration and nothing else: the program to put behind `rocprofv3 --kernel-trace --`
when the per-launch timeline of a frame is wanted (tools/timeline_from_trace.py reads the trace).
SCENE / NX / NY / NS / STRIDE / FIRST select the frame, RT_OPTS=key=value,... the knobs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import accelerated_ray_tracer_amd as art
art.init(0)
for kv in filter(None, os.environ.get("RT_OPTS", "").split(",")):
    k, v = kv.split("="); art.set_option(k, int(v))
scene, nx, ny, ns = os.environ.get("SCENE", "random_scene"), int(os.environ.get("NX", "1200")), int(os.environ.get("NY", "800")), int(os.environ.get("NS", "500"))
stride, first = int(os.environ.get("STRIDE", "1")), int(os.environ.get("FIRST", "0"))
img, iw, ih = art.default_texture(scene)
hs = art.HostScene(scene, nx, ny, img, iw, ih)
ds = art.DeviceScene(hs)
f = hs.frame(nx=nx, ny=ny, ns=ns, tile_rows=4 if stride > 1 else ny, tile_first=first, tile_stride=stride)
rows = art.rt_lib().rt_frame_local_rows(f)
buf = torch.zeros((rows, nx, 3), dtype=torch.float32, device="cuda")
for _ in range(int(os.environ.get("FRAMES", "3"))):
    _, st = ds.render(f, out=buf.data_ptr(), blocking=True)
print(f"{scene} {nx}x{ny}@{ns} 1/{stride}: {st.ms_render:.3f} ms, {st.rays} rays, heavy {st.reserved}", flush=True)

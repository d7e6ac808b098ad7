#!/usr/bin/env python3
"""How long does the most expensive part of the frame take when it has the GPU to itself?  Renders single 8-row bands
(1200 x 8 pixels = 150 waves on 1024 SIMDs) of the headline frame at the given spp: the slowest band's time is a lower
bound for ANY partition of the frame over any number of GPUs, because a pixel's samples are one sequential chain."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import accelerated_ray_tracer_amd as art
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 500
opts = dict(kv.split("=") for kv in sys.argv[2:])
art.init(0)
for k, v in opts.items(): art.set_option(k, int(v))
hs = art.HostScene("random_scene", 1200, 800)
ds = art.DeviceScene(hs)
buf = torch.zeros((800, 1200, 3), dtype=torch.float32, device="cuda")
_, st = ds.render(hs.frame(ns=ns), out=buf.data_ptr(), blocking=True)
_, st = ds.render(hs.frame(ns=ns), out=buf.data_ptr(), blocking=True)
print(f"whole frame @ {ns} spp: {st.ms_render:.2f} ms, {st.rays} rays  opts {opts}")
res = []
for band in range(0, 100, 3):
    _, sb = ds.render(hs.frame(ns=ns, tile_rows=8, tile_first=band, tile_stride=10**6), out=buf.data_ptr(), blocking=True)
    res.append((sb.ms_render, band, sb.rays, sb.reserved, sb.workgroups))
res.sort(reverse=True)
for ms, band, rays, heavy, wgs in res[:6]: print(f"  rows {band*8:3d}..{band*8+7:3d}: {ms:8.2f} ms   {rays/(1200*8*ns):5.2f} rays/sample   heavy pixels {heavy}  workgroups {wgs}")
print(f"  median band {np.median([r[0] for r in res]):.2f} ms")

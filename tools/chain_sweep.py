#!/usr/bin/env python3
"""Time of the worst 8-row band (rows 408..415) alone on the GPU under different scheduler settings."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import accelerated_ray_tracer_amd as art
ns = int(sys.argv[1]); cfgs = sys.argv[2:]
art.init(0)
hs = art.HostScene("random_scene", 1200, 800); ds = art.DeviceScene(hs)
buf = torch.zeros((800, 1200, 3), dtype=torch.float32, device="cuda")
for c in cfgs:
    art.reset_options()
    for kv in c.split(","):
        if kv: k, v = kv.split("="); art.set_option(k, int(v))
    best = 1e9
    for _ in range(2):
        _, sb = ds.render(hs.frame(ns=ns, tile_rows=8, tile_first=51, tile_stride=10**6), out=buf.data_ptr(), blocking=True); best = min(best, sb.ms_render)
    print(f"{c:70s} band 408..415: {best:8.2f} ms", flush=True)

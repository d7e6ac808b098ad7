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
hs = art.HostScene("random_scene", 1200, 800)
ds = art.DeviceScene(hs)
for stride in [int(x) for x in (sys.argv[1:] or ["1", "2", "4", "8"])]:
    worst = 0.0
    for first in range(stride):
        f = hs.frame(ns=500, tile_rows=4 if stride > 1 else 800, tile_first=first, tile_stride=stride)
        rows = art.rt_lib().rt_frame_local_rows(f)
        buf = torch.zeros((rows, 1200, 3), dtype=torch.float32, device="cuda")
        ts = []
        for _ in range(3):
            _, st = ds.render(f, out=buf.data_ptr(), blocking=True)
            ts.append(st.ms_render)
        worst = max(worst, min(ts))
        print(f"stride {stride} rank {first}: {min(ts):8.3f} ms  {st.rays/1e6:8.1f} Mrays  heavy {st.reserved}", flush=True)
    print(f"== N={stride}: slowest rank {worst:.3f} ms", flush=True)

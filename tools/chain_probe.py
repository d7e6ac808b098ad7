#!/usr/bin/env python3
"""Renders only rows 408..415 of the headline frame (its dearest pixels; 13.8 M rays, a trivial amount of work): the time
is the sequential chain of the dearest pixel through the three parts of the cost-aware schedule, with the GPU otherwise
empty.  RT_OPTS=key=value,... sets knobs; with RT_LIB_OVERRIDE=<diag build> it also prints the trace time per ray of the
tier loops (s_memtime ticks).  Round-1 result: 84 ms with tier 0 (auto), 173 ms without -- profiles/r01i_chain_probe.log."""
import os, sys
sys.path.insert(0, '/root/repo')
import torch, accelerated_ray_tracer_amd as art
art.init(0)
for kv in filter(None, os.environ.get("RT_OPTS", "").split(",")):
    k, v = kv.split("="); art.set_option(k, int(v))
hs = art.HostScene("random_scene", 1200, 800)
ds = art.DeviceScene(hs)
f = hs.frame(ns=500, tile_rows=8, tile_first=51, tile_stride=100)   # rows 408..415: the dearest pixels of the frame
rows = art.rt_lib().rt_frame_local_rows(f)
buf = torch.zeros((rows, 1200, 3), dtype=torch.float32, device="cuda")
ts=[]
for _ in range(3):
    _, st = ds.render(f, out=buf.data_ptr(), blocking=True); ts.append(st.ms_render)
print(os.environ.get("RT_OPTS",""), "rows", rows, "min ms %.3f" % min(ts), "rays %.1f M" % (st.rays/1e6), "heavy", st.reserved, "wgs", st.workgroups)
if os.environ.get("RT_LIB_OVERRIDE"):
    import ctypes as C, numpy as np
    c = np.zeros(16, np.uint64)
    L = art.rt_lib(); L.rt_debug_counters.argtypes = [C.c_void_p, C.c_void_p]
    L.rt_debug_counters(ds._p, c.ctypes.data)
    n = max(int(c[15]), 1)
    print("diag (thread 0 of tier workgroups, last frame): tier rays %d  trace ticks/ray %.0f (s_memtime)" % (n, int(c[14]) / n))

#!/usr/bin/env python3
"""Strong-scaling estimate on ONE GPU: time the row partition each of N ranks would render (tile_rows = 4, rank r takes
tiles r, r+N, ...) and compare with the whole frame.  efficiency ~= T1 / (N * max_r T_N(r)); the RCCL gather
(1.44 MB per rank at N=8) is not included."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import accelerated_ray_tracer_amd as art
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 100
opts = dict(kv.split("=") for kv in sys.argv[2:])
art.init(0)
for k, v in opts.items(): art.set_option(k, int(v))
hs = art.HostScene("random_scene", 1200, 800)
ds = art.DeviceScene(hs)
buf = torch.zeros((800, 1200, 3), dtype=torch.float32, device="cuda")
def timed(frame, reps=3):
    best = 1e9; st = None
    for _ in range(reps):
        _, st = ds.render(frame, out=buf.data_ptr(), blocking=True); best = min(best, st.ms_render)
    return best, st
t1, st1 = timed(hs.frame(ns=ns))
print(f"N=1: {t1:.2f} ms  rays {st1.rays}  wg {st1.workgroups}  opts {opts}")
for n in (2, 4, 8):
    ts = []
    for r in range(n):
        t, st = timed(hs.frame(ns=ns, tile_rows=4, tile_first=r, tile_stride=n), reps=2)
        ts.append(t)
    print(f"N={n}: per-rank ms min {min(ts):.2f} max {max(ts):.2f} mean {np.mean(ts):.2f}  wg {st.workgroups}  -> efficiency ~ {t1/(n*max(ts)):.3f}")

#!/usr/bin/env python3
"""When do the waves of a frame's last launch end?  (diagnostic build, RT_LIB_OVERRIDE=.../librt_mi355x_diag.so)
Prints, per millisecond after the first wave's start, how many ordinary and how many initially-sparse (tier) waves ended:
a long thin tail after the bulk means the frame is bound by a few pixels' sequential chains, not by throughput.
Usage: [SCENE= NX= NY=] diag_wave_ends.py ns [key=value ...]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import accelerated_ray_tracer_amd as art
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 500
scene, nx, ny = os.environ.get("SCENE", "random_scene"), int(os.environ.get("NX", "1200")), int(os.environ.get("NY", "800"))
opts = dict(kv.split("=") for kv in sys.argv[2:])
art.init(0)
for k, v in opts.items(): art.set_option(k, int(v))
img, iw, ih = art.default_texture(scene)
hs = art.HostScene(scene, nx, ny, img, iw, ih)
ds = art.DeviceScene(hs)
stride, first = int(os.environ.get("STRIDE", "1")), int(os.environ.get("FIRST", "0"))   # one rank's share of a STRIDE-GPU run
fkw = dict(nx=nx, ny=ny, ns=ns, tile_rows=4 if stride > 1 else ny, tile_first=first, tile_stride=stride)
fb, st = ds.render(hs.frame(**fkw))
fb, st = ds.render(hs.frame(**fkw))
BINS = 192
h = np.zeros(2 * BINS, np.uint64)
L = art.rt_lib(); L.rt_debug_wave_ends.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
assert L.rt_debug_wave_ends(ds._p, h.ctypes.data, 2 * BINS) == 0
a, b = h[:BINS].astype(int), h[BINS:].astype(int)
print(f"scene {scene} {nx}x{ny}@{ns} opts {opts}: frame {st.ms_render:.2f} ms (diag build), heavy {st.reserved}, wgs {st.workgroups} x {st.threads_per_group}; waves ended per ms of the last launch (ordinary | started sparse):")
tot = a.sum() + b.sum(); run = 0
for ms in range(BINS):
    if a[ms] or b[ms]:
        run += a[ms] + b[ms]
        print(f"  {ms:4d} ms  {a[ms]:6d} | {b[ms]:6d}   cumulative {100.0 * run / tot:6.2f} %")
# the waves that ended last: what was the lane that ran out of work last doing?
n_waves = st.workgroups * (st.threads_per_group // 64)
w = np.zeros(2 * n_waves, np.uint64)
L.rt_debug_wave_last.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
assert L.rt_debug_wave_last(ds._p, w.ctypes.data, n_waves) == 0
rows = []
for k in range(n_waves):
    a0, a1 = int(w[2 * k]), int(w[2 * k + 1])
    if a0 == 0: continue
    rows.append((a0 >> 32, a0 & 0xFFFFFFFF, a1 >> 60, (a1 >> 59) & 1, a1 & 0xFFFFFFF, k))
rows.sort(reverse=True)
import oracle   # (experiment tooling: what the pixels that end the launch cost in all)
orc = oracle.OracleScene("bouncing" if scene == "random_scene" else scene, nx, ny, img, iw, ih)
Lo = oracle.lib()
Lo.orc_row_pixel_rays.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_ulonglong, C.c_int, C.c_void_p]
rows_global = art.local_rows_to_global(hs.frame(**fkw))
bg = np.ascontiguousarray(orc.background, np.float32)
row_cache = {}
def pixel_rays(i, lrow):
    j = int(rows_global[lrow])
    if j not in row_cache:
        out = np.zeros(nx, np.uint64)
        Lo.orc_row_pixel_rays(orc.h, nx, ny, ns, bg.ctypes.data, orc.gradient, 1984, j, out.ctypes.data)
        row_cache[j] = out
    return int(row_cache[j][i]), j
mean = st.rays / (len(rows_global) * nx)
print(f"mean rays per pixel of this share: {mean:.0f}")
print("last 16 waves: done_ms  last pixel fetched at ms  (duration)  queue  started-sparse  pixel(i,global row)  rays of that pixel (x mean)  us per ray")
for d_us, f_us, src, sp, pix, k in rows[:16]:
    r, j = pixel_rays(pix % nx, pix // nx)
    print(f"  wave {k:5d}  done {d_us/1000:7.2f}  fetched {f_us/1000:7.2f}  ({(d_us-f_us)/1000:6.2f} ms)  queue {src}  sparse {sp}  pixel ({pix % nx}, {j})  {r} rays ({r / mean:.2f} x)  {(d_us - f_us) / max(r * (ns - 32) / ns, 1):.1f} us/ray")

#!/usr/bin/env python3
"""Per-phase wall-clock shares of the wavefront kernel (diagnostic build, RT_LIB_OVERRIDE=...diag.so).  Shares only; never quote its run time."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import accelerated_ray_tracer_amd as art
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 20
opts = dict(kv.split("=") for kv in sys.argv[2:])
art.init(0)
art.set_option("kernel", 4)
for k, v in opts.items(): art.set_option(k, int(v))
hs = art.HostScene("random_scene", 1200, 800)
ds = art.DeviceScene(hs)
fb, st = ds.render(hs.frame(ns=ns))
c = np.zeros(16, np.uint64)
L = art.rt_lib(); L.rt_debug_counters.argtypes = [C.c_void_p, C.c_void_p]
L.rt_debug_counters(ds._p, c.ctypes.data)
t = [int(x) for x in c[:8]]; waves = int(c[8]); tot = sum(t)
names = ["T phase", "barrier after T", "C phase", "barrier after C", "D phase", "barrier after D", "E phase", "barrier+bookkeeping"]
print(f"variant {st.kernel_variant} wg {st.workgroups} x {st.threads_per_group}  rays {st.rays}  ms {st.ms_render:.2f}  waves {waves}  opts {opts}")
for n, v in zip(names, t): print(f"  {n:22s} {100.0*v/max(tot,1):6.2f} %   ({v/max(waves,1)/100.0:10.0f} x100 ticks per wave)")

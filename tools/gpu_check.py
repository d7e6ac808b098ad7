#!/usr/bin/env python3
"""Quick GPU-vs-oracle comparison over all scenes and both kernels (development aid; tests/ holds the real checks)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import accelerated_ray_tracer_amd as art
import oracle

cases = [("two_spheres", 200, 100, 4), ("bouncing", 96, 64, 8), ("book1", 96, 64, 8), ("cornell", 64, 64, 8), ("simple_light", 64, 32, 8), ("original", 48, 48, 4), ("perlin", 48, 24, 4), ("earth", 48, 24, 4),
         ("cornell_smoke", 64, 64, 8), ("final", 64, 64, 8)]
art.init(0)
for name, nx, ny, ns in cases:
    img, iw, ih = art.default_texture(name)
    hs = art.HostScene(name, nx, ny, img, iw, ih)
    orc = oracle.OracleScene(name, nx, ny, img, iw, ih)
    ref, cnt = orc.render(ns)
    ds = art.DeviceScene(hs)
    for kernel in (0, 3, 4):
        art.set_option("kernel", kernel)
        t = time.time()
        fb, st = ds.render(hs.frame(ns=ns))
        dt = time.time() - t
        diff = np.abs(fb - ref)
        nbad = int((fb.view(np.uint32) != ref.view(np.uint32)).any(axis=2).sum())
        print(f"{name:14s} kernel {kernel} variant {st.kernel_variant:5d} rays gpu {st.rays} oracle {cnt['rays']}  max|d| {diff.max():.3e} "
              f"pixels differing {nbad}/{nx*ny}  nan {int(np.isnan(fb).sum())}  {st.ms_render:.2f} ms (wall {dt*1e3:.0f})", flush=True)
    ds.close()

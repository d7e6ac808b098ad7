#!/usr/bin/env python3
"""Renders every reference scene at the size its host function uses (src/main.cu:654-1305) and saves 8-bit PNGs +
4x4 box-averaged float arrays under gpurun_out/views/, for comparison with the README illustrations the reference
holds in images/ (tests/golden/reference_images.npz).  GPU only.  Usage: tools/render_reference_views.py [spp]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.realpath(__file__))))
import accelerated_ray_tracer_amd as art

def main():
    spp_arg = sys.argv[1] if len(sys.argv) > 1 else "256"
    names = sys.argv[2].split(",") if len(sys.argv) > 2 else ["bouncing", "final", "cornell", "cornell_smoke", "original", "perlin",
                                                               "quads", "simple_light", "checker", "earth", "book1"]
    out = os.path.join("gpurun_out", "views"); os.makedirs(out, exist_ok=True)
    art.init(0)
    from PIL import Image
    for name in names:
        img, iw, ih = art.default_texture(name)
        hs = art.HostScene(name, 0, 0, img, iw, ih)
        ds = art.DeviceScene(hs)
        spp = hs.ns if spp_arg == "default" else int(spp_arg)   # "default": the ns of the reference host function
        t = time.time()
        fb, st = ds.render(hs.frame(ns=spp))
        dt = time.time() - t
        px = np.clip((fb[::-1] * np.float32(255.99)).astype(np.int32), 0, 255).astype(np.uint8)   # int(255.99f*c), main.cu:722
        Image.fromarray(px).save(os.path.join(out, f"{name}.png"))
        print(f"{name}: {hs.nx}x{hs.ny} @ {spp} spp, {st.rays/1e6:.0f} Mrays in {dt*1e3:.0f} ms", flush=True)
        ds.close() if hasattr(ds, "close") else None

if __name__ == "__main__":
    main()

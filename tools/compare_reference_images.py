#!/usr/bin/env python3
"""Compares full frames rendered by tools/render_reference_views.py (at the reference host functions' own nx, ny, ns,
seed 1984) with the README illustrations the reference holds under images/ -- the only outputs of the real CUDA
binary that exist.  Runs in the build container only (reads /root/reference/images).  The PNGs are lossless
8-bit copies of the PPM the reference printed (int(255.99f*c), main.cu:715-727), so "exact" below means the same
8-bit triple.  Usage: tools/compare_reference_images.py [views_dir]..."""
import sys
import numpy as np
from PIL import Image

# scene name -> README image produced by the reference's current code for that scene (see SURVEY.md "images/ <-> code")
PAIRS = [("quads", "quads"), ("checker", "checkered"), ("earth", "textureWrap"), ("perlin", "perlin"),
         ("simple_light", "poolBall"), ("bouncing", "utk"), ("cornell", "redBlue"), ("original", "alfredo2"),
         ("final", "finalScene")]

def main():
    dirs = sys.argv[1:] or ["gpurun_out/views"]
    for d in dirs:
        print(f"== {d}")
        print(f"{'scene':13s} {'reference image':16s} {'size':>9s} {'exact':>9s} {'|d|<=1':>9s} {'|d|<=2':>9s} {'max':>4s} {'mean|d|':>8s}")
        for mine, ref in PAIRS:
            a = np.asarray(Image.open(f"{d}/{mine}.png").convert("RGB")).astype(int)
            b = np.asarray(Image.open(f"/root/reference/images/{ref}.png").convert("RGB")).astype(int)
            diff = np.abs(a - b); m = diff.max(2)
            print(f"{mine:13s} {ref + '.png':16s} {a.shape[1]:4d}x{a.shape[0]:<4d} {100*(m==0).mean():8.3f}% {100*(m<=1).mean():8.3f}% "
                  f"{100*(m<=2).mean():8.3f}% {diff.max():4d} {diff.mean():8.4f}")

if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Rays per pixel (world->hit calls) on the headline frame, from the GPU ray counter run one row at a time is too slow,
so this uses the CPU oracle on a subsampled set of rows: per-pixel ray counts decide the strong-scaling floor, because a
pixel's samples are one sequential random stream."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ctypes as C
import oracle
nx, ny, ns = 1200, 800, int(sys.argv[1]) if len(sys.argv) > 1 else 16
sc = oracle.OracleScene("bouncing", nx, ny)
L = oracle.lib()
rows = list(range(0, ny, 8))
per_pixel = []
bg = np.zeros(3, np.float32)
fb = np.zeros((ny, nx, 3), np.float32)
# one call per row gives per-row counts; per-pixel needs one call per pixel: approximate with per-row then sample pixels in the heaviest rows
row_rays = {}
for j in rows:
    cnt = np.zeros(8, np.uint64)
    L.orc_render(sc.h, fb.ctypes.data, nx, ny, ns, 2.2, bg.ctypes.data, 0, 1984, j, j + 1, cnt.ctypes.data, 1)
    row_rays[j] = int(cnt[0]) / (nx * ns)
print("rays per sample by row (every 8th row, bottom to top):")
print(" ".join(f"{row_rays[j]:.2f}" for j in rows))
print(f"frame mean ~ {np.mean(list(row_rays.values())):.3f}")

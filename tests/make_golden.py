#!/usr/bin/env python3
"""Regenerates tests/golden/oracle_frames.npz: small fp32 frames rendered by the oracle (oracle/rt_oracle.cpp).
These are the oracle's own outputs (the reference cannot run here); they pin the checker and give the GPU tests
fixtures that need no oracle build.  Run from the repo root:  python tests/make_golden.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
CASES = [("two_spheres", 40, 20, 2), ("bouncing", 48, 32, 4), ("book1", 48, 32, 2), ("cornell", 32, 32, 4),
         ("cornell_smoke", 32, 32, 4), ("final", 32, 32, 2), ("checker", 40, 20, 2), ("perlin", 32, 16, 2), ("quads", 40, 20, 2),
         ("degenerate", 32, 16, 4)]
# (scenes with image textures are covered against the live oracle in test_gpu_parity.py, with assets/*.ppm)
out = {}
for name, nx, ny, ns in CASES:
    fb, _ = oracle.OracleScene(name, nx, ny).render(ns)
    out[f"{name}_{nx}_{ny}_{ns}"] = fb
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_frames.npz"), **out)
print({k: v.shape for k, v in out.items()})

#!/usr/bin/env python3
"""Builds tests/golden/reference_image_pins.npz from the output images the reference itself holds.

The reference has no tests or golden vectors, but its README illustrations (/root/reference/images/*.png) are
lossless 8-bit copies of the PPM its CUDA binary printed (int(255.99f*c), src/main.cu:715-727) for the scene
functions of the current source at their own nx, ny, ns and seed 1984.  They are the only outputs of the real
reference that exist, so they are what the oracle and the HIP path are pinned against.

Only data is taken (decoded pixel values); nothing from the reference's sources.  To keep the fixture small it holds
  rows_<scene>  uint8 [local_rows][nx][3]: the rows of the row partition tile_rows=4, tile_first=0, tile_stride=8
                (one tile in eight), in framebuffer order (row 0 = bottom of the image, as main.cu:115 indexes fb)
  box_<scene>   float32 [ny/8][nx/8][3]: 8x8 box means of the whole image, framebuffer order
Run in the build container (the GPU box has no /root/reference):  python tests/golden/make_reference_image_pins.py
"""
import json
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.realpath(__file__))
REF = "/root/reference/images"
# scene name (host/rtw_scenes.cpp) -> README image rendered by the reference's current code for that scene
PAIRS = {"quads": "quads.png", "checker": "checkered.png", "earth": "textureWrap.png", "perlin": "perlin.png",
         "simple_light": "poolBall.png", "bouncing": "utk.png", "cornell": "redBlue.png", "original": "alfredo2.png",
         "final": "finalScene.png"}
TILE_ROWS, TILE_FIRST, TILE_STRIDE = 4, 0, 8


def pinned_rows(ny):
    return [r for r in range(ny) if (r // TILE_ROWS) % TILE_STRIDE == TILE_FIRST]


def main():
    out, meta = {}, {"tile_rows": TILE_ROWS, "tile_first": TILE_FIRST, "tile_stride": TILE_STRIDE, "images": {}}
    for scene, png in PAIRS.items():
        img = np.asarray(Image.open(os.path.join(REF, png)).convert("RGB"))   # top row first
        fb = img[::-1]                                                          # framebuffer order
        ny, nx, _ = fb.shape
        out["rows_" + scene] = np.ascontiguousarray(fb[pinned_rows(ny)])
        out["box_" + scene] = fb[: ny // 8 * 8, : nx // 8 * 8].astype(np.float32).reshape(ny // 8, 8, nx // 8, 8, 3).mean((1, 3))
        meta["images"][scene] = {"file": "images/" + png, "nx": nx, "ny": ny}
    np.savez_compressed(os.path.join(HERE, "reference_image_pins.npz"), **out)
    with open(os.path.join(HERE, "reference_image_pins.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()

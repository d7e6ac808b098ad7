#!/usr/bin/env python3
"""Decode the reference's textures/*.jpg (data assets, not source) to binary PPMs under assets/.

The reference decodes them at run time with the stb_image.h it vendors (src/image_io.h:26).  JPEG decoders differ by
a level or two per texel (Pillow/libjpeg vs stb: 1.3 % of earthmap's texels, 92 % of poolball's), which shows up when
pixels are compared with the reference's own output images, so the assets are made with the reference's decoder:
oracle/decode_texture.c compiles that header from where it lies under /root/reference (make -C oracle
_ref/decode_texture).  Run once, in the build container; /root/reference does not exist on the GPU box."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
tool = os.path.join(ROOT, "oracle", "_ref", "decode_texture")
subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "_ref/decode_texture"])
for name in (sys.argv[1:] or ["earthmap", "poolball", "8ball"]):
    subprocess.check_call([tool, f"/root/reference/textures/{name}.jpg", os.path.join(ROOT, "assets", name + ".ppm")])

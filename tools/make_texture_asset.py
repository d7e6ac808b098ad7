#!/usr/bin/env python3
"""Decode the reference's textures/earthmap.jpg (a data asset, not source) to a binary PPM under assets/.

The reference decodes it with stb_image at run time (src/image_io.h:24-41); this container has Pillow (libjpeg),
whose IDCT may differ from stb's by +-1 per channel, so earth-textured pixels carry that tolerance against a real
reference run.  Run once, here; /root/reference does not exist on the GPU box."""
import sys
from PIL import Image
src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/textures/earthmap.jpg"
dst = sys.argv[2] if len(sys.argv) > 2 else "assets/earthmap.ppm"
im = Image.open(src).convert("RGB")
with open(dst, "wb") as f:
    f.write(b"P6\n%d %d\n255\n" % im.size)
    f.write(im.tobytes())
print(dst, im.size)

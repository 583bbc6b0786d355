#!/usr/bin/env python3
"""Decode the reference's earth texture into a binary PPM (P6) data asset.

The reference loads textures/earthmap.jpg with the Rust `image` crate
(src/ray_tracing/materials/texture/image.rs:15-43).  This image has no JPEG
library headers for C++, so the asset is decoded once here (PIL / libjpeg) and
committed as raw RGB8.  Decoders may differ from the reference's zune-jpeg by
+-1 LSB per channel.

Usage (authoring container only, /root/reference is not on the GPU box):
    python tools/make_earthmap_ppm.py /root/reference/textures/earthmap.jpg \
        raytracing_rust_amd/assets/earthmap.ppm
"""
import sys
from PIL import Image

src, dst = sys.argv[1], sys.argv[2]
im = Image.open(src).convert("RGB")
w, h = im.size
with open(dst, "wb") as f:
    f.write(b"P6\n%d %d\n255\n" % (w, h))
    f.write(im.tobytes())
print(f"wrote {dst}: {w}x{h}")

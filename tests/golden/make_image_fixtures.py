"""Block means of the reference's own output images: the only reference-produced data for this path.
reference_image_blocks.npz: images/cornell_box.png, images/output_cornell_smoke.png (and images/quick_test.png cropped)
averaged over a 30 x 30 grid of blocks (20 x 20 pixels at 600 x 600) per channel, float32.
reference_image_blocks10.npz: the final scene's images - images/quick_test.png (400 x 400, the `quick` entry point),
images/output_final_scene_800x800_5000spp.png and images/output_final_scene_800x800_10000spp.png - over a 10 x 10 grid:
that scene's boxes, sphere cluster and marble are random per run (final_scene.rs:58,155-159), so only its large-scale
layout and brightness are comparable.  The tests render the same scene, average the same way and compare loosely
(the reference's sample count and random streams are unknown).

    python tests/golden/make_image_fixtures.py /root/reference
"""
import os
import sys

import numpy as np
from PIL import Image

GRID = 30
IMAGES = {"cornell": "cornell_box.png", "smoke": "output_cornell_smoke.png", "quick": "quick_test.png"}


def block_means(rgb, grid=GRID):
    h, w, _ = rgb.shape
    assert h % grid == 0 and w % grid == 0, (h, w)
    return rgb.reshape(grid, h // grid, grid, w // grid, 3).astype(np.float64).mean(axis=(1, 3)).astype(np.float32)


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    out = {}
    for name, fn in IMAGES.items():
        img = np.asarray(Image.open(os.path.join(ref, "images", fn)).convert("RGB"))
        if img.shape[0] % GRID:                                   # quick_test.png is 400 x 400: crop to 390 x 390
            n = img.shape[0] // GRID * GRID
            img = img[:n, :n]
        out[name] = block_means(img)
        print(name, img.shape, out[name].mean(axis=(0, 1)))
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_image_blocks.npz"), **out)


FINAL_IMAGES = {"quick": "quick_test.png", "final_5000spp": "output_final_scene_800x800_5000spp.png",
                "final_10000spp": "output_final_scene_800x800_10000spp.png"}


def main10():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    out = {}
    for name, fn in FINAL_IMAGES.items():
        img = np.asarray(Image.open(os.path.join(ref, "images", fn)).convert("RGB"))
        out[name] = block_means(img, 10)
        print(name, img.shape, out[name].mean(axis=(0, 1)))
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_image_blocks10.npz"), **out)


if __name__ == "__main__":
    main()
    main10()

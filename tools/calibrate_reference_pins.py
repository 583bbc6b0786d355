#!/usr/bin/env python3
"""How far apart are renders of the final scene that differ only in their random SCENE content (box heights
final_scene.rs:58, cluster spheres 155-159, Perlin tables procedural/noise.rs:22-29)?  The reference's generator is
unseeded, so its checked-in images show one unknown draw; the tolerances of tests/test_reference_images.py for
`quick` and `final` are set from this spread.  Uses the CPU oracle; prints per pair the three statistics the test
checks on 10 x 10 block means of the 8-bit image: largest channel difference of the global mean, mean absolute block
difference, correlation.

    python tools/calibrate_reference_pins.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import helpers as H  # noqa: E402
from raytracing_rust_amd import scenes  # noqa: E402

GRID = 10


def render(name, width, spp, depth, seed):
    g = H.oracle_graph()
    w, l, cam = scenes.build(name, g, width=width, spp=spp, depth=depth, prefix="orc_", lib=H.oracle_lib(), scene_seed=seed)
    acc, _ = H.oracle_render(g, w, l, cam)
    rgb = np.zeros((cam.image_height, width, 3), dtype=np.uint8)
    H.oracle_lib().orc_resolve_rgb8(H.dptr(acc), width, cam.image_height, spp, rgb.ctypes.data_as(H.PU8))
    n = width // GRID
    return rgb.reshape(GRID, n, GRID, n, 3).astype(np.float64).mean(axis=(1, 3))


def stats(a, b):
    return (np.abs(a.mean(axis=(0, 1)) - b.mean(axis=(0, 1))).max(), np.abs(a - b).mean(), np.corrcoef(a.ravel(), b.ravel())[0, 1])


def main():
    ref = np.load(os.path.join(ROOT, "tests", "golden", "reference_image_blocks10.npz"))
    seeds = [scenes.SCENE_SEED, 11, 22, 33, 44]
    for name, cfg, keys in (("quick", (400, 100, 20), ["quick"]), ("final", (200, 400, 75), ["final_10000spp", "final_5000spp"])):
        bs = [render(name, *cfg, seed) for seed in seeds]
        worst = [0.0, 0.0, 1.0]
        for i in range(len(bs)):
            for j in range(i):
                s = stats(bs[i], bs[j])
                worst = [max(worst[0], s[0]), max(worst[1], s[1]), min(worst[2], s[2])]
        print(f"{name} {cfg}: own renders, worst pair over {len(seeds)} scene seeds: mean {worst[0]:.2f}, blocks {worst[1]:.2f}, corr {worst[2]:.4f}")
        for key in keys:
            r = ref[key].astype(np.float64)
            for seed, b in zip(seeds, bs):
                s = stats(b, r)
                print(f"   seed {seed:#x} vs {key}: mean {s[0]:.2f}, blocks {s[1]:.2f}, corr {s[2]:.4f}; after scaling to the image's mean: "
                      f"blocks {np.abs(b * r.mean() / b.mean() - r).mean():.2f}")


if __name__ == "__main__":
    main()

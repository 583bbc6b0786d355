"""Renders against the reference's own output images (the only reference-produced data for this path).

tests/golden/reference_image_blocks.npz holds 30 x 30 block means of images/cornell_box.png and
images/output_cornell_smoke.png (made by tests/golden/make_image_fixtures.py).  The reference's sample count and
random streams are unknown, and its caustics are resolution dependent, so the comparison is loose and stated here:
per-channel global mean within 1.5 of 255, mean absolute block difference below 4 of 255, correlation of the block
means above 0.99.  (A wrong light size, wall colour, medium density or gamma fails these by a wide margin.)"""
import os

import numpy as np
import pytest

import raytracing_rust_amd as rt
from raytracing_rust_amd import scenes
import helpers as H

BLOCKS = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_image_blocks.npz"))
GRID = 30


def check(name, rgb):
    n = rgb.shape[0] // GRID
    mine = rgb.reshape(GRID, n, GRID, n, 3).astype(np.float64).mean(axis=(1, 3))
    ref = BLOCKS[name].astype(np.float64)
    assert np.abs(mine.mean(axis=(0, 1)) - ref.mean(axis=(0, 1))).max() < 1.5, (mine.mean(axis=(0, 1)), ref.mean(axis=(0, 1)))
    assert np.abs(mine - ref).mean() < 4.0, np.abs(mine - ref).mean()
    assert np.corrcoef(mine.ravel(), ref.ravel())[0, 1] > 0.99


@pytest.mark.parametrize("name", ["cornell", "smoke"])
def test_oracle_matches_reference_image(name):
    """The CPU oracle at a quarter of the reference's resolution (150 x 150, 100 spp, depth 50, resolved to 8 bits by
    the oracle's color_to_rgb_with_samples)."""
    g = H.oracle_graph()
    w, l, cam = scenes.build(name, g, width=150, spp=100, depth=50, prefix="orc_", lib=H.oracle_lib())
    acc, _ = H.oracle_render(g, w, l, cam)
    rgb = np.zeros((150, 150, 3), dtype=np.uint8)
    H.oracle_lib().orc_resolve_rgb8(H.dptr(acc), 150, 150, 100, rgb.ctypes.data_as(H.PU8))
    check(name, rgb)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cornell", "smoke"])
def test_gpu_render_matches_reference_image(gpu_ctx, name):
    """The HIP path at the reference's resolution (600 x 600, 144 spp, depth 50) through the C ABI."""
    g = rt.Graph()
    w, l, cam = scenes.build(name, g, width=600, spp=144, depth=50)
    scene = rt.Scene.from_graph(gpu_ctx, g, w, l)
    acc = scene.render(cam, rt.render_opts(cam, seed=scenes.PATH_SEED))
    check(name, gpu_ctx.resolve_rgb8(acc, 144))
    scene.close()

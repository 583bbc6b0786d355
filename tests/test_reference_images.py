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


# ------------------------------------------------------------------ the final scene (`quick`, `final`) ---
# The final scene's boxes, sphere cluster and Perlin tables are drawn from an unseeded generator in the reference
# (final_scene.rs:58,155-159, procedural/noise.rs:22-29), so its checked-in images show one unknown draw and only the
# large-scale layout and brightness are comparable: 10 x 10 block means (tests/golden/reference_image_blocks10.npz).
# Tolerances, stated here, are 1.5 x the worst pair among this build's own renders over five scene seeds
# (tools/calibrate_reference_pins.py: global mean 2.8 of 255, mean block difference 5.4, correlation 0.958): global mean
# per channel within 4.0 of 255, mean absolute block difference below 8.0, correlation above 0.94.  Measured against
# images/quick_test.png: 0.2 to 2.8, 3.2 to 5.0, 0.969 to 0.989 over those seeds, against
# images/output_final_scene_800x800_10000spp.png: 0.3 to 1.5, 4.3 to 6.1, 0.962 to 0.987 - the reference's images look
# like one more scene seed.  images/output_final_scene_800x800_5000spp.png is the exception: it is 20 of 255 brighter
# than the reference's other two images of the same scene (which agree with each other to 1.0 in the global mean), so
# the checked-in code cannot have produced all three; it is compared for layout only (correlation above 0.94).
BLOCKS10 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_image_blocks10.npz"))


def check10(key, rgb, layout_only=False):
    n = rgb.shape[0] // 10
    mine = rgb.reshape(10, n, 10, n, 3).astype(np.float64).mean(axis=(1, 3))
    ref = BLOCKS10[key].astype(np.float64)
    assert np.corrcoef(mine.ravel(), ref.ravel())[0, 1] > 0.94, np.corrcoef(mine.ravel(), ref.ravel())[0, 1]
    if layout_only:
        return
    assert np.abs(mine.mean(axis=(0, 1)) - ref.mean(axis=(0, 1))).max() < 4.0, (mine.mean(axis=(0, 1)), ref.mean(axis=(0, 1)))
    assert np.abs(mine - ref).mean() < 8.0, np.abs(mine - ref).mean()


def _oracle_rgb(name, width, spp, depth):
    g = H.oracle_graph()
    w, l, cam = scenes.build(name, g, width=width, spp=spp, depth=depth, prefix="orc_", lib=H.oracle_lib())
    acc, _ = H.oracle_render(g, w, l, cam)
    rgb = np.zeros((cam.image_height, width, 3), dtype=np.uint8)
    H.oracle_lib().orc_resolve_rgb8(H.dptr(acc), width, cam.image_height, spp, rgb.ctypes.data_as(H.PU8))
    return rgb


def test_oracle_matches_reference_quick_image():
    """The oracle on the `quick` entry point exactly as main.rs:31-39 runs it (400 x 400, 100 spp, depth 20) against
    images/quick_test.png."""
    check10("quick", _oracle_rgb("quick", 400, 100, 20))


def test_oracle_matches_reference_final_images():
    """The oracle on the final scene at a quarter of the resolution (200 x 200, 400 spp, depth 75) against the
    reference's 800 x 800 images."""
    rgb = _oracle_rgb("final", 200, 400, 75)
    check10("final_10000spp", rgb)
    check10("final_5000spp", rgb, layout_only=True)


@pytest.mark.gpu
def test_gpu_render_matches_reference_quick_image(gpu_ctx):
    """The HIP path on `quick` at full size through the C ABI."""
    g = rt.Graph()
    w, l, cam = scenes.build("quick", g)
    scene = rt.Scene.from_graph(gpu_ctx, g, w, l)
    acc = scene.render(cam, rt.render_opts(cam, seed=scenes.PATH_SEED))
    check10("quick", gpu_ctx.resolve_rgb8(acc, cam.samples_per_pixel))
    scene.close()


@pytest.mark.gpu
def test_gpu_render_matches_reference_final_images(gpu_ctx):
    """The HIP path on `final` exactly as main.rs:22-29 runs it - 800 x 800, 5000 spp (4900 rendered, divided by 5000:
    camera.rs:95,317), depth 75: the configuration the benchmark times - against the reference's 800 x 800 images."""
    g = rt.Graph()
    w, l, cam = scenes.build("final", g)
    assert (cam.image_width, cam.samples_per_pixel, cam.sqrt_spp, cam.max_depth) == (800, 5000, 70, 75)
    scene = rt.Scene.from_graph(gpu_ctx, g, w, l)
    acc = scene.render(cam, rt.render_opts(cam, seed=scenes.PATH_SEED))
    rgb = gpu_ctx.resolve_rgb8(acc, cam.samples_per_pixel)
    check10("final_10000spp", rgb)
    check10("final_5000spp", rgb, layout_only=True)
    scene.close()

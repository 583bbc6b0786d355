"""Frozen first-hit vectors (tests/golden/*.npz, written by tests/golden/make_golden.py from the oracle; the
reference itself has no fixtures for this path).  The oracle, the host build of the device core and, on a GPU, the
kernels must all reproduce them bit for bit."""
import glob
import os

import numpy as np
import pytest

import raytracing_rust_amd as rt
from raytracing_rust_amd import scenes
import helpers as H

HERE = os.path.dirname(os.path.abspath(__file__))
FILES = sorted(glob.glob(os.path.join(HERE, "golden", "first_hit_*.npz")))


def load(path):
    z = np.load(path, allow_pickle=False)
    name, width = os.path.basename(path)[len("first_hit_"):-len(".npz")].rsplit("_", 1)
    assert int(z["scene_seed"]) == scenes.SCENE_SEED
    return name, int(width), z


def test_fixtures_exist():
    assert len(FILES) == 3


@pytest.mark.parametrize("path", FILES, ids=os.path.basename)
def test_oracle_reproduces_golden(path):
    name, width, z = load(path)
    g = H.oracle_graph()
    w, l, cam = scenes.build(name, g, width=width, prefix="orc_", lib=H.oracle_lib())
    H.assert_aov_equal(H.oracle_first_hit(g, w, z["rays"]), z, f"oracle vs golden {name}")


@pytest.mark.parametrize("path", FILES, ids=os.path.basename)
def test_device_core_reproduces_golden(path):
    name, width, z = load(path)
    g = rt.Graph()
    w, l, cam = scenes.build(name, g, width=width)
    H.assert_aov_equal(H.hostsim_first_hit(rt.flatten(g, w, l), z["rays"]), z, f"hostsim vs golden {name}")


@pytest.mark.gpu
@pytest.mark.parametrize("path", FILES, ids=os.path.basename)
def test_gpu_reproduces_golden(path, gpu_ctx):
    name, width, z = load(path)
    g = rt.Graph()
    w, l, cam = scenes.build(name, g, width=width)
    scene = rt.Scene.from_graph(gpu_ctx, g, w, l)
    try:
        H.assert_aov_equal(scene.first_hit(z["rays"]), z, f"GPU vs golden {name}")
    finally:
        scene.close()

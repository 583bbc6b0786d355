#!/usr/bin/env python3
"""Writes the golden first-hit vectors under tests/golden/.

The reference has no tests, fixtures or seeded outputs for this path and cannot be built here (no Rust toolchain),
so these vectors are NOT outputs of the reference: they are the oracle's answers (oracle/oracle.cpp, the line-by-line
CPU restatement) frozen at the time of writing, after the oracle passed its closed-form known-answer tests.  They
guard the oracle, the host layer's scene builders and the kernels against silent drift: any of the three changing a
single bit of t, normal, ids or the front flag on these rays fails tests/test_golden.py.

    python tests/golden/make_golden.py          # rewrites the .npz files
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import raytracing_rust_amd as rt  # noqa: E402
from raytracing_rust_amd import scenes  # noqa: E402
import helpers as H  # noqa: E402

CASES = [("cornell", 40, 0.0), ("smoke", 40, 0.0), ("final", 48, 0.37)]


def main():
    for name, width, time in CASES:
        g = H.oracle_graph()
        w, l, cam = scenes.build(name, g, width=width, prefix="orc_", lib=H.oracle_lib())
        rays = rt.pixel_center_rays(cam, time=time)
        h = H.oracle_first_hit(g, w, rays)
        np.savez_compressed(os.path.join(HERE, f"first_hit_{name}_{width}.npz"), rays=rays, t=h["t"], normal=h["normal"],
                            prim_id=h["prim_id"], mat_id=h["mat_id"], front=h["front"],
                            scene_seed=np.uint64(scenes.SCENE_SEED))
        print(name, width, "hits", int(np.isfinite(h["t"]).sum()), "of", len(rays))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""How much does ray order matter to the traversal?  Secondary rays of the final scene are traced by the first-hit
entry point in several orders; read the k_first_hit durations from a rocprofv3 kernel trace (dispatch order =
print order).  Tuning tool, not part of the product."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracing_rust_amd as rt  # noqa: E402


def morton3(q):
    def spread(v):
        v = v.astype(np.uint64) & 0x3FF
        v = (v | (v << 16)) & 0x30000FF
        v = (v | (v << 8)) & 0x300F00F
        v = (v | (v << 4)) & 0x30C30C3
        v = (v | (v << 2)) & 0x9249249
        return v
    return spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)


def main():
    ctx = rt.Context(0)
    g = rt.Graph()
    world, lights, cam = rt.scenes.build("final", g, width=800, spp=100, depth=75)
    scene = rt.Scene.from_graph(ctx, g, world, lights)
    rays = rt.pixel_center_rays(cam, time=0.3).reshape(cam.image_height, cam.image_width, 7)
    tiles = rays.reshape(cam.image_height // 8, 8, cam.image_width // 8, 8, 7).transpose(0, 2, 1, 3, 4).reshape(-1, 7)
    hit = scene.first_hit(tiles)
    ok = np.isfinite(hit["t"])
    rng = np.random.default_rng(7)
    p = tiles[:, 0:3] + tiles[:, 3:6] * np.where(ok, hit["t"], 1.0)[:, None]
    d = hit["normal"] + rng.normal(size=(len(tiles), 3)) * 0.8
    sec = np.concatenate([p, d, tiles[:, 6:7]], axis=1)[ok]
    n = len(sec)
    lo, hi = sec[:, 0:3].min(0), sec[:, 0:3].max(0)
    q = ((sec[:, 0:3] - lo) / (hi - lo + 1e-9) * 1023).astype(np.uint32)
    mort = morton3(q)
    octant = ((sec[:, 3] < 0).astype(np.uint64) | ((sec[:, 4] < 0).astype(np.uint64) << 1) | ((sec[:, 5] < 0).astype(np.uint64) << 2))
    orders = {
        "tile order": np.arange(n),
        "shuffled": rng.permutation(n),
        "octant, morton30": np.argsort((octant << 30) | mort, kind="stable"),
        "morton30, octant": np.argsort((mort << 3) | octant, kind="stable"),
        "octant, morton15": np.argsort((octant << 15) | (mort >> 15), kind="stable"),
        "octant only": np.argsort(octant, kind="stable"),
    }
    ref = None
    for name, o in orders.items():
        r = np.ascontiguousarray(sec[o])
        t0 = time.perf_counter()
        h = scene.first_hit(r)
        dt = time.perf_counter() - t0
        inv = np.empty(n, dtype=np.int64); inv[o] = np.arange(n)
        t = h["t"][inv]
        if ref is None:
            ref = t
        assert np.array_equal(t, ref)
        print(f"{name:20s} n={n} call {dt * 1e3:.1f} ms", flush=True)


if __name__ == "__main__":
    main()

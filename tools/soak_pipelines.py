#!/usr/bin/env python3
"""Soak of the identical-paths property: the wavefront pipeline (streamed sections, box trees walked nearer child
first, sorted shading) against the fused pipeline (one lane per pixel, the reference's visiting order) over several
path seeds on the final scene, a recipe full of coincident quads and spheres, and the stress scene.  Both must count
the same ray segments and light probes and produce the same image up to the order of the f64 additions.

    python tools/soak_pipelines.py        (on an MI355X; exit status 1 on any difference)
"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import raytracing_rust_amd as rt
from raytracing_rust_amd import _capi, scenes
import recipes
ctx = rt.Context(0)
bad = 0
def check(name, scene, cam, seeds):
    global bad
    for seed in seeds:
        out = {}
        for pipe in (_capi.PIPELINE_FUSED, _capi.PIPELINE_WAVEFRONT):
            ctx.reset_stats()
            out[pipe] = (scene.render(cam, rt.render_opts(cam, seed=seed, pipeline=pipe)), ctx.stats())
        (a, sa), (b, sb) = out[_capi.PIPELINE_FUSED], out[_capi.PIPELINE_WAVEFRONT]
        ok = sa["segments"] == sb["segments"] and sa["pdf_probes"] == sb["pdf_probes"] and np.allclose(b, a, rtol=1e-11, atol=1e-13)
        print(name, seed, "ok" if ok else "DIFF", sa["segments"], sb["segments"], float(np.abs(a - b).max()), flush=True)
        bad += 0 if ok else 1
g = rt.Graph(); w, l, cam = scenes.build("final", g, width=200, spp=16, depth=50)
sc = rt.Scene.from_graph(ctx, g, w, l); check("final", sc, cam, range(100, 108)); sc.close()
for mode in ("1", "2"):
    os.environ["RTX_WF_NEAR"] = mode
    g = rt.Graph(); w, _, _ = recipes.ties_in_box_trees(g)
    cam = rt.initialize_camera(rt.camera_desc(aspect_ratio=1.5, image_width=150, samples_per_pixel=16, max_depth=20, background=(0.5, 0.6, 0.8), vfov=50.0, lookfrom=(60, 120, 520), lookat=(0, 40, 0)))
    sc = rt.Scene.from_graph(ctx, g, w, None); check("ties near=" + mode, sc, cam, range(200, 206)); sc.close()
os.environ["RTX_WF_NEAR"] = "1"
g = rt.Graph(); w, l, cam = scenes.build("stress", g, width=320, spp=4, depth=20, n_prims=300000)
sc = rt.Scene.from_graph(ctx, g, w, l); check("stress", sc, cam, range(300, 304)); sc.close()
print("bad", bad)
sys.exit(1 if bad else 0)

"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle on the same seeded inputs."""
import numpy as np
import pytest

import raytracing_rust_amd as rt
from raytracing_rust_amd import _capi, scenes
import helpers as H
import recipes

pytestmark = pytest.mark.gpu


def _pair(name, **kw):
    go, gp = H.oracle_graph(), rt.Graph()
    wo, lo, cam_o = scenes.build(name, go, prefix="orc_", lib=H.oracle_lib(), **kw)
    wp, lp, cam = scenes.build(name, gp, **kw)
    assert bytes(cam_o) == bytes(cam)
    return go, wo, lo, gp, wp, lp, cam


@pytest.mark.parametrize("name,width", [("cornell", 600), ("smoke", 300), ("final", 800), ("quick", 400)])
def test_first_hit_aov_bit_exact_named_scenes(gpu_ctx, name, width):
    """t, normal, primitive id, material id and front flag on every pixel-centre ray, at BASELINE sizes."""
    go, wo, lo, gp, wp, lp, cam = _pair(name, width=width)
    scene = rt.Scene.from_graph(gpu_ctx, gp, wp, lp)
    rays = rt.pixel_center_rays(cam, time=0.37)
    H.assert_aov_equal(H.oracle_first_hit(go, wo, rays), scene.first_hit(rays), name)
    scene.close()


def test_first_hit_aov_bit_exact_stress_scene(gpu_ctx):
    """The synthetic stress configuration at 200 000 primitives (a 13 MB hit program, walked from global memory):
    first hits bit-exact against the oracle's recursive BVH, and both pipelines trace the same paths through it (the
    wavefront pipeline walks this BVH, its head section, nearer child first; the fused one in the reference's order)."""
    go, wo, lo, gp, wp, lp, cam = _pair("stress", width=480, spp=4, depth=12, n_prims=200000)
    scene = rt.Scene.from_graph(gpu_ctx, gp, wp, lp)
    rays = rt.pixel_center_rays(cam, time=0.5)
    ref = H.oracle_first_hit(go, wo, rays)
    H.assert_aov_equal(ref, scene.first_hit(rays), "stress")
    assert np.isfinite(ref["t"]).sum() > 20000
    out = {}
    for pipe in (_capi.PIPELINE_FUSED, _capi.PIPELINE_WAVEFRONT):
        gpu_ctx.reset_stats()
        out[pipe] = (scene.render(cam, rt.render_opts(cam, seed=9, pipeline=pipe)), gpu_ctx.stats())
    (a, sa), (b, sb) = out[_capi.PIPELINE_FUSED], out[_capi.PIPELINE_WAVEFRONT]
    assert sa["segments"] == sb["segments"] and sa["pdf_probes"] == sb["pdf_probes"] and sb["head_launches"] > 0
    np.testing.assert_allclose(b, a, rtol=1e-11, atol=1e-13)
    scene.close()


def test_first_hit_aov_bit_exact_stress_scene_at_full_size(gpu_ctx):
    """The synthetic stress configuration AT SIZE: 1 M spheres and quads in one BvhNode (a 70 MB hit program), the
    3840 x 2160 camera, first hits on the pixel centres of every eighth pixel in both directions (480 x 270 rays)
    bit-exact against the oracle's recursive BVH.  The product's tree is built on the GPU (rtx_bvh_leaf_order), the
    oracle's by the recursion of bvh.rs:26-71."""
    go, gp = H.oracle_graph(), rt.Graph()
    gp.use_context(gpu_ctx)
    wo, lo, cam_o = scenes.build("stress", go, prefix="orc_", lib=H.oracle_lib())
    wp, lp, cam = scenes.build("stress", gp)
    assert bytes(cam_o) == bytes(cam) and (cam.image_width, cam.image_height) == (3840, 2160)
    scene = rt.Scene.from_graph(gpu_ctx, gp, wp, lp)
    rays = rt.pixel_center_rays(cam, time=0.5).reshape(2160, 3840, 7)[4::8, 4::8].reshape(-1, 7)
    assert rays.shape[0] == 480 * 270
    ref = H.oracle_first_hit(go, wo, rays)
    H.assert_aov_equal(ref, scene.first_hit(rays), "stress 1M")
    assert np.isfinite(ref["t"]).sum() > 50000
    scene.close()


@pytest.mark.parametrize("recipe", recipes.ALL, ids=lambda r: r.__name__)
def test_first_hit_aov_bit_exact_recipes(gpu_ctx, recipe):
    """Random and edge-case graphs: nested transforms, BVHs of mixed leaves, ties, axis-parallel rays, empty lists."""
    go, gp = H.oracle_graph(), rt.Graph()
    wo, _, rays = recipe(go)
    wp, _, _ = recipe(gp)
    scene = rt.Scene.from_graph(gpu_ctx, gp, wp, None)
    H.assert_aov_equal(H.oracle_first_hit(go, wo, rays), scene.first_hit(rays), recipe.__name__)
    scene.close()


@pytest.mark.parametrize("name,width,spp,depth", [("cornell", 48, 16, 12), ("smoke", 96, 16, 20), ("quick", 96, 16, 20),
                                                  ("final", 96, 16, 75)])
def test_gpu_matches_host_build_of_same_code(gpu_ctx, name, width, spp, depth):
    """Same counter-based RNG addressing on both sides: the GPU render and the host build of the device core trace
    the same paths; only libm-vs-OCML transcendentals differ, so sums agree to a tight tolerance (stated here: the
    segment counts within 0.2 %, at least 99.5 % of the channels within 1e-6 relative).  The host build is bit-exact
    against the oracle on textures, light pdfs and first hits (CPU tier), so this pins, on the GPU and through the
    default wavefront pipeline, everything the final scene has and the Cornell box lacks: both BVHs with their head
    and deferred sections, Perlin noise (procedural/noise.rs:35-127), the image lookup (image.rs:58-88), metal
    (metal.rs:24-38), the moving sphere and both constant media (constant_medium.rs:42-90).  Beyond the block means of
    the reference's PNGs the oracle itself is parity unpinned (no toolchain, no vectors in the reference)."""
    gp = rt.Graph()
    wp, lp, cam = scenes.build(name, gp, width=width, spp=spp, depth=depth)
    desc = rt.flatten(gp, wp, lp)
    scene = rt.Scene(gpu_ctx, desc)
    gpu_ctx.reset_stats()
    a = scene.render(cam, rt.render_opts(cam, seed=11))
    st = gpu_ctx.stats()
    b, cb = H.hostsim_render(desc, cam, rt.render_opts(cam, seed=11))
    assert st["paths"] == cb["paths"] == width * cam.image_height * cam.sqrt_spp ** 2
    assert abs(st["segments"] - cb["segments"]) <= 0.002 * cb["segments"]
    close = np.isclose(a, b, rtol=1e-6, atol=1e-9)
    assert close.mean() > 0.995, f"{name}: only {close.mean():.4f} of the channels agree"
    assert b.sum() > 0 and abs(a.sum() - b.sum()) < 2e-3 * b.sum()
    scene.close()


def test_gpu_matches_host_build_on_a_scene_walked_from_global_memory(gpu_ctx):
    """The same comparison on the stress scene at 200 000 primitives: its BVH does not fit LDS, so the GPU's default
    pipeline walks it over the folded table (single-precision culling, exact leaf boxes, every lane a step per trip)
    while the host build walks the program itself in the reference's order.  Same paths, hence the same tolerance; 64
    light quads make every diffuse bounce evaluate 64 light pdfs (hittable_list.rs:78-97)."""
    gp = rt.Graph()
    wp, lp, cam = scenes.build("stress", gp, width=96, spp=16, depth=20, n_prims=200000)
    desc = rt.flatten(gp, wp, lp)
    scene = rt.Scene(gpu_ctx, desc)
    gpu_ctx.reset_stats()
    a = scene.render(cam, rt.render_opts(cam, seed=12))
    st = gpu_ctx.stats()
    b, cb = H.hostsim_render(desc, cam, rt.render_opts(cam, seed=12))
    assert st["paths"] == cb["paths"] == 96 * cam.image_height * cam.sqrt_spp ** 2 and st["head_launches"] > 0
    assert abs(st["segments"] - cb["segments"]) <= 0.002 * cb["segments"]
    close = np.isclose(a, b, rtol=1e-6, atol=1e-9)
    assert close.mean() > 0.995, f"stress: only {close.mean():.4f} of the channels agree"
    assert b.sum() > 0 and abs(a.sum() - b.sum()) < 2e-3 * b.sum()
    scene.close()


@pytest.mark.parametrize("depth,defocus", [(2, 0.0), (12, 0.0), (12, 2.0)])
def test_material_probe_matches_host_build(gpu_ctx, depth, defocus):
    """Metal with and without fuzz, glass, Lambertian walls, a moving sphere, a box-bounded and a sphere-bounded medium
    under a quad light and a sphere light (two lights: the index draw of hittable_list.rs:94 and Sphere::random run),
    at depth 2 (every path is one scatter and one more segment) and depth 12, and through a lens (defocus angle 2
    degrees, camera.rs:139-164).  The host build of the same code traces the same paths (same RNG addressing); it is
    compared with the oracle value by value under replayed random numbers on this very scene (CPU tier,
    test_replay_cpu.py), so this carries that comparison onto the device.  Tolerance as in the test above."""
    gp = rt.Graph()
    wp, lp, camkw = recipes.material_probe(gp)
    camkw.update(max_depth=depth, image_width=96, defocus_angle=defocus, focus_dist=100.0 if defocus else 10.0)
    cam = rt.initialize_camera(rt.camera_desc(**camkw))
    desc = rt.flatten(gp, wp, lp)
    scene = rt.Scene(gpu_ctx, desc)
    for pipe in (_capi.PIPELINE_WAVEFRONT, _capi.PIPELINE_FUSED):
        gpu_ctx.reset_stats()
        a = scene.render(cam, rt.render_opts(cam, seed=13, pipeline=pipe))
        st = gpu_ctx.stats()
        b, cb = H.hostsim_render(desc, cam, rt.render_opts(cam, seed=13))
        assert st["paths"] == cb["paths"] == 96 * 96 * 16
        assert abs(st["segments"] - cb["segments"]) <= 0.002 * cb["segments"] and cb["segments"] > 1.5 * cb["paths"]
        assert abs(st["pdf_probes"] - cb["pdf_probes"]) <= 0.002 * cb["pdf_probes"] and cb["pdf_probes"] > 0
        close = np.isclose(a, b, rtol=1e-6, atol=1e-9)
        assert close.mean() > 0.995, f"only {close.mean():.4f} of the channels agree"
        assert b.sum() > 0 and abs(a.sum() - b.sum()) < 2e-3 * b.sum()
    scene.close()


def test_stress_scene_render_path_at_full_size(gpu_ctx):
    """The render path of the stress configuration AT SIZE: 1 M spheres and quads in one BvhNode, 64 lights.  The
    default (wavefront) pipeline walks that BVH as a stream over the folded table, nearer child first, from global
    memory; the fused pipeline walks the program in the reference's order.  480 x 270 pixels of the 16:9 camera, 4
    samples, depth 12: the same segments and light probes, radiance within 1e-11 (order of the f64 additions)."""
    gp = rt.Graph()
    gp.use_context(gpu_ctx)
    wp, lp, cam = scenes.build("stress", gp, width=480, spp=4, depth=12)
    assert (cam.image_width, cam.image_height) == (480, 270)
    desc = rt.flatten(gp, wp, lp)
    assert desc.n_spheres + desc.n_quads == 1000000 and desc.n_lights == 64
    scene = rt.Scene(gpu_ctx, desc)
    out = {}
    for pipe in (_capi.PIPELINE_FUSED, _capi.PIPELINE_WAVEFRONT):
        gpu_ctx.reset_stats()
        out[pipe] = (scene.render(cam, rt.render_opts(cam, seed=19, pipeline=pipe)), gpu_ctx.stats())
    (a, sa), (b, sb) = out[_capi.PIPELINE_FUSED], out[_capi.PIPELINE_WAVEFRONT]
    assert sa["paths"] == sb["paths"] == 480 * 270 * 4
    assert sa["segments"] == sb["segments"] and sa["pdf_probes"] == sb["pdf_probes"] and sb["head_launches"] > 0
    assert sb["segments"] > 1.5 * sb["paths"]
    np.testing.assert_allclose(b, a, rtol=1e-11, atol=1e-13)
    assert a.sum() > 0
    scene.close()


def test_thin_media_drawn_first_trace_the_same_paths(gpu_ctx, monkeypatch):
    """RTX_MEDIA_DRAW_FIRST=1 compiles the final scene's fog (radius 5000, density 1e-4) so that it draws its free path
    before it intersects its boundary and leaves when the path is longer than the ray can be inside
    (constant_medium.rs:60-78 returns false then whatever the crossings are).  Same paths as the default form, in both
    pipelines."""
    gp = rt.Graph()
    wp, lp, cam = scenes.build("final", gp, width=160, spp=16, depth=40)
    desc = rt.flatten(gp, wp, lp)
    plain = rt.Scene(gpu_ctx, desc)
    monkeypatch.setenv("RTX_MEDIA_DRAW_FIRST", "1")
    first = rt.Scene(gpu_ctx, desc)
    monkeypatch.delenv("RTX_MEDIA_DRAW_FIRST")
    for pipe in (_capi.PIPELINE_WAVEFRONT, _capi.PIPELINE_FUSED):
        out = []
        for scene in (plain, first):
            gpu_ctx.reset_stats()
            out.append((scene.render(cam, rt.render_opts(cam, seed=83, pipeline=pipe)), gpu_ctx.stats()))
        (a, sa), (b, sb) = out
        assert sa["segments"] == sb["segments"] and sa["pdf_probes"] == sb["pdf_probes"] and sa["segments"] > 2 * sa["paths"]
        np.testing.assert_allclose(b, a, rtol=1e-11, atol=1e-13)
    plain.close(); first.close()


def test_texture_probe_first_bounce_matches_oracle(gpu_ctx):
    """Checker (checker.rs:41-56), a checker of a checker with noise and image children, marble noise at two scales
    (texture/noise.rs:29-35), the earth image, a missing image (cyan) - each as the emission of a light, rendered with
    max_depth 1 so that every pixel holds exactly the texture value at its camera ray's hit.  Expected values: the
    oracle's hit record (u, v, p) on the same rays fed to the oracle's Texture::value.  Tolerance 1e-12 absolute on
    the quads (their (u, v) are rational in the ray: no transcendental before the lookup; sin in the noise texture
    differs by an ulp between libm and the device); on the two spheres (u, v) come from acos / atan2, so a lookup
    next to a texel or checker boundary may land on the other side: at least 99.5 % of their pixels must agree."""
    go, gp = H.oracle_graph(), rt.Graph()
    wo, camkw, tex_o = recipes.texture_probe(go)
    wp, _, _ = recipes.texture_probe(gp)
    cam = rt.initialize_camera(rt.camera_desc(**camkw))
    scene = rt.Scene.from_graph(gpu_ctx, gp, wp, None)
    seen = np.zeros(len(tex_o), dtype=np.int64)
    for s in range(cam.sqrt_spp ** 2):
        want, hit = H.oracle_first_bounce_emission(go, wo, tex_o, cam, s, 4242)
        got = scene.render(cam, rt.render_opts(cam, sample_begin=s, sample_end=s + 1, seed=4242))
        err = np.abs(got - want).max(axis=2).ravel()
        prim = hit["prim_id"]
        flat = (prim < 6) | (prim == 0xFFFFFFFF)           # quads, and misses (background)
        assert err[flat].max() <= 1e-12, f"sample {s}: quad pixels off by {err[flat].max()}"
        for k in (6, 7):
            sel = prim == k
            assert (err[sel] <= 1e-12).mean() >= 0.995, f"sample {s}: sphere {k}: {(err[sel] > 1e-12).sum()} of {sel.sum()} pixels differ"
        seen += np.bincount(prim[prim < len(tex_o)], minlength=len(tex_o))
    assert (seen > 2000).all(), seen                       # every texture was looked at
    scene.close()


def test_converged_image_full_size_quick(gpu_ctx):
    """The `quick` entry point at its full size (main.rs:31-39: final scene, 400 x 400, 100 spp, depth 20) against the
    oracle at equal spp with independent random streams.  Stated tolerances, on radiance clamped to the display range
    [0, 1] (color.rs:44-49 clamps the output there; the light's 7.0 would otherwise dominate every norm):
    per-pixel relative RMSE (RMSE / mean) below 0.50 (measured: 0.43 to 0.44) and within 6 % of the same distance between two GPU renders with
    different seeds (two independent estimates of one image are exactly that far apart); the same on 8 x 8 block
    means within 15 % of the GPU-to-GPU distance (a systematic 2 % error of the block means would add 10 %); mean
    radiance within 0.5 %."""
    go, wo, lo, gp, wp, lp, cam = _pair("quick")
    assert (cam.image_width, cam.image_height, cam.sqrt_spp, cam.max_depth) == (400, 400, 10, 20)
    scene = rt.Scene.from_graph(gpu_ctx, gp, wp, lp)
    n = cam.sqrt_spp ** 2
    a1 = scene.render(cam, rt.render_opts(cam, seed=301)) / n
    a2 = scene.render(cam, rt.render_opts(cam, seed=302)) / n
    b, _ = H.oracle_render(go, wo, lo, cam, seed=303)
    b /= n
    assert np.isfinite(a1).all() and np.isfinite(b).all()

    def rrmse(x, y, k):
        x, y = np.minimum(x, 1.0), np.minimum(y, 1.0)
        if k > 1:
            x, y = (z.reshape(400 // k, k, 400 // k, k, 3).mean(axis=(1, 3)) for z in (x, y))
        return np.sqrt(((x - y) ** 2).mean()) / y.mean()

    noise1, noise8 = rrmse(a1, a2, 1), rrmse(a1, a2, 8)
    for a in (a1, a2):
        d1, d8 = rrmse(a, b, 1), rrmse(a, b, 8)
        assert d1 < 0.50 and abs(d1 / noise1 - 1.0) < 0.06, f"per pixel: GPU-vs-oracle {d1:.4f}, GPU-vs-GPU {noise1:.4f}"
        assert d8 < 1.15 * noise8, f"8x8 blocks: GPU-vs-oracle {d8:.4f}, GPU-vs-GPU {noise8:.4f}"
    rel = abs(0.5 * (a1 + a2).mean() - b.mean()) / b.mean()
    assert rel < 0.005, f"mean radiance differs by {rel:.4f}"
    scene.close()


@pytest.mark.parametrize("name,spp,depth", [("cornell", 400, 50), ("smoke", 400, 50), ("quick", 144, 20)])
def test_render_matches_oracle_statistically(gpu_ctx, name, spp, depth):
    """Converged-image check at equal spp with independent seeds, calibrated by the Monte Carlo noise itself: the
    8x8-block means of the GPU render must be as close to the oracle's as two GPU renders with different seeds are
    to each other (factor 1.5 + 0.5 % slack), and the global mean radiance must agree within 1.5 %.
    Tolerance: relative L1 distance of block means; no bit-exactness is claimed for shaded radiance."""
    W = 64
    go, wo, lo, gp, wp, lp, cam = _pair(name, width=W, spp=spp, depth=depth)
    scene = rt.Scene.from_graph(gpu_ctx, gp, wp, lp)
    n = cam.sqrt_spp ** 2
    a1 = scene.render(cam, rt.render_opts(cam, seed=101)) / n
    a2 = scene.render(cam, rt.render_opts(cam, seed=202)) / n
    b, _ = H.oracle_render(go, wo, lo, cam)
    b /= n
    assert not np.isnan(a1).any()
    blk = lambda x: x.reshape(W // 8, 8, W // 8, 8, 3).mean(axis=(1, 3))
    dist = lambda x, y: np.abs(blk(x) - blk(y)).mean() / blk(y).mean()
    noise = dist(a1, a2)
    assert dist(a1, b) < 1.5 * noise + 0.005, f"{name}: GPU-vs-oracle {dist(a1, b):.4f}, GPU-vs-GPU noise {noise:.4f}"
    assert dist(a2, b) < 1.5 * noise + 0.005, f"{name}: GPU-vs-oracle {dist(a2, b):.4f}, GPU-vs-GPU noise {noise:.4f}"
    rel_mean = abs(0.5 * (a1 + a2).mean() - b.mean()) / b.mean()
    assert rel_mean < 0.015, f"{name}: mean radiance differs by {rel_mean:.4f}"
    scene.close()


@pytest.mark.parametrize("name,width,spp", [("final", 200, 16), ("smoke", 120, 16), ("cornell", 120, 16)])
def test_wavefront_and_fused_pipelines_trace_the_same_paths(gpu_ctx, name, width, spp):
    """Both pipelines address the RNG by (pixel, sample, bounce), so they trace identical paths: the accumulators
    agree up to the order of the f64 additions (tolerance 1e-11 relative, stated here), and so do the segment and
    light-probe counters.  Covers what only the wavefront path has: queues, compaction, deferred sections, the tail
    kernel."""
    gp = rt.Graph()
    wp, lp, cam = scenes.build(name, gp, width=width, spp=spp, depth=40)
    scene = rt.Scene.from_graph(gpu_ctx, gp, wp, lp)
    out = {}
    for pipe in (_capi.PIPELINE_FUSED, _capi.PIPELINE_WAVEFRONT):
        gpu_ctx.reset_stats()
        out[pipe] = (scene.render(cam, rt.render_opts(cam, seed=77, pipeline=pipe)), gpu_ctx.stats())
    (a, sa), (b, sb) = out[_capi.PIPELINE_FUSED], out[_capi.PIPELINE_WAVEFRONT]
    assert sa["paths"] == sb["paths"] == cam.image_width * cam.image_height * cam.sqrt_spp ** 2
    assert sa["segments"] == sb["segments"] and sa["pdf_probes"] == sb["pdf_probes"]
    np.testing.assert_allclose(b, a, rtol=1e-11, atol=1e-13)
    assert a.sum() > 0
    scene.close()


@pytest.mark.parametrize("depth", [1, 2, 5])
def test_shallow_paths_leave_dead_entries_that_change_nothing(gpu_ctx, depth):
    """The wavefront shading kernel reserves a queue slot for every hit that may go on; a path that ends after all
    (here mostly: its last allowed bounce, camera.rs:174-176) leaves a dead entry that the traversal kernels walk and
    the next sort drops.  With depths this small most reserved slots end dead: counters and image must still be
    those of the fused pipeline, and dead entries must not be counted as segments."""
    gp = rt.Graph()
    wp, lp, cam = scenes.build("final", gp, width=120, spp=9, depth=depth)
    scene = rt.Scene.from_graph(gpu_ctx, gp, wp, lp)
    out = {}
    for pipe in (_capi.PIPELINE_FUSED, _capi.PIPELINE_WAVEFRONT):
        gpu_ctx.reset_stats()
        out[pipe] = (scene.render(cam, rt.render_opts(cam, seed=5, pipeline=pipe)), gpu_ctx.stats())
    (a, sa), (b, sb) = out[_capi.PIPELINE_FUSED], out[_capi.PIPELINE_WAVEFRONT]
    assert sa["paths"] == sb["paths"] == 120 * 120 * 9
    assert sa["segments"] == sb["segments"] and sa["pdf_probes"] == sb["pdf_probes"]
    assert sb["segments"] <= sb["paths"] * depth
    np.testing.assert_allclose(b, a, rtol=1e-11, atol=1e-13)
    scene.close()


@pytest.mark.parametrize("mode", ["0", "1"])
def test_streamed_sections_trace_the_same_paths(gpu_ctx, monkeypatch, mode):
    """The head section of the world program (the ground BVH of the final scene) and the deferred section (its sphere
    cluster) can be walked per ray or as streams with refilled lanes (RTX_WF_STREAM=0 / 1; streams are the
    default).  Either way the wavefront pipeline must trace the fused pipeline's paths."""
    gp = rt.Graph()
    wp, lp, cam = scenes.build("final", gp, width=160, spp=16, depth=40)
    scene = rt.Scene.from_graph(gpu_ctx, gp, wp, lp)
    gpu_ctx.reset_stats()
    a = scene.render(cam, rt.render_opts(cam, seed=78, pipeline=_capi.PIPELINE_FUSED))
    sa = gpu_ctx.stats()
    monkeypatch.setenv("RTX_WF_STREAM", mode)
    gpu_ctx.reset_stats()
    b = scene.render(cam, rt.render_opts(cam, seed=78, pipeline=_capi.PIPELINE_WAVEFRONT))
    sb = gpu_ctx.stats()
    assert sa["segments"] == sb["segments"] and sa["pdf_probes"] == sb["pdf_probes"]
    assert (sb["head_launches"] > 0) == (mode == "1")
    np.testing.assert_allclose(b, a, rtol=1e-11, atol=1e-13)
    scene.close()


def test_sections_are_streamed_by_call_size(gpu_ctx, monkeypatch):
    """By default a section that lives in LDS is walked as a stream only when a pipeline's share of the call is large
    (RTX_WF_STREAM_MIN camera paths, 16 M): a short call is launch latency, and a stream is one more launch per
    iteration.  Either way the paths are the same."""
    gp = rt.Graph()
    wp, lp, cam = scenes.build("final", gp, width=160, spp=16, depth=40)
    scene = rt.Scene.from_graph(gpu_ctx, gp, wp, lp)
    out = []
    for stream_min in ("100000", None):
        if stream_min is None:
            monkeypatch.delenv("RTX_WF_STREAM_MIN")
        else:
            monkeypatch.setenv("RTX_WF_STREAM_MIN", stream_min)
        gpu_ctx.reset_stats()
        out.append((scene.render(cam, rt.render_opts(cam, seed=79, pipeline=_capi.PIPELINE_WAVEFRONT)), gpu_ctx.stats()))
    (a, sa), (b, sb) = out
    assert sa["head_launches"] > 0 and sb["head_launches"] == 0      # 410 k paths: streamed only with the threshold lowered
    assert sa["segments"] == sb["segments"] and sa["pdf_probes"] == sb["pdf_probes"]
    np.testing.assert_allclose(b, a, rtol=1e-11, atol=1e-13)
    scene.close()


@pytest.mark.parametrize("near", ["0", "1", "2"])
def test_sections_nearer_child_first_trace_the_same_paths(gpu_ctx, monkeypatch, near):
    """The ground of the final scene (head section) and its sphere cluster (deferred section) are binary trees of boxes
    over BOX6 records and sphere runs; their streams can walk the nearer child of every box first (RTX_WF_NEAR=0: the
    reference's left-to-right order in both; 1, the default: nearer first in the deferred section, and in a head section
    too large for LDS; 2: in both).  The closest
    hit does not depend on the visiting order (ties are resolved as the fixed order resolves them), so all must trace
    the fused pipeline's paths, which walks in the reference's order."""
    gp = rt.Graph()
    wp, lp, cam = scenes.build("final", gp, width=240, spp=16, depth=30)
    scene = rt.Scene.from_graph(gpu_ctx, gp, wp, lp)
    gpu_ctx.reset_stats()
    a = scene.render(cam, rt.render_opts(cam, seed=90, pipeline=_capi.PIPELINE_FUSED))
    sa = gpu_ctx.stats()
    monkeypatch.setenv("RTX_WF_NEAR", near)
    monkeypatch.setenv("RTX_WF_STREAM", "1")   # a call this small would not stream its sections by itself
    gpu_ctx.reset_stats()
    b = scene.render(cam, rt.render_opts(cam, seed=90, pipeline=_capi.PIPELINE_WAVEFRONT))
    sb = gpu_ctx.stats()
    assert sa["segments"] == sb["segments"] and sa["pdf_probes"] == sb["pdf_probes"] and sb["tail_launches"] > 0
    np.testing.assert_allclose(b, a, rtol=1e-11, atol=1e-13)
    scene.close()


@pytest.mark.parametrize("recipe,near", [(recipes.large_bvh_beyond_lds, None), (recipes.empty_lists_and_media, None),
                                         (recipes.nested_transforms, None), (recipes.nested_transforms, "per-ray"),
                                         (recipes.ties_in_box_trees, None),
                                         (recipes.ties_in_box_trees, "2"), (recipes.ties_on_box_boundaries, None)],
                         ids=lambda v: getattr(v, "__name__", v))
def test_pipelines_agree_on_recipes(gpu_ctx, monkeypatch, recipe, near):
    """The same identical-paths check on graphs the shipped scenes do not have: a program too large for LDS (global
    memory walk, deferred box BVH), generic (box-bounded, transformed) media next to sphere-bounded ones, nested
    transforms, box trees full of coincident quads and spheres (walked nearer child first in the deferred section,
    and with RTX_WF_NEAR=2 in the head section as well).  The frame is 97 x 61 so that the 8 x 8 tiles are clipped on
    both edges.  "per-ray" leaves the sections to the kernel that walks the whole program for each ray: with transformed
    leaves in the head section that is the general walk, not the sweep without transform contexts."""
    if near == "per-ray":
        monkeypatch.setenv("RTX_WF_STREAM", "0")
    else:
        if near is not None:
            monkeypatch.setenv("RTX_WF_NEAR", near)
        monkeypatch.setenv("RTX_WF_STREAM", "1")   # sections as streams (a call this small would walk them per ray by itself)
    gp = rt.Graph()
    wp, _, _ = recipe(gp)
    cam = rt.initialize_camera(rt.camera_desc(aspect_ratio=97.0 / 61.0, image_width=97, samples_per_pixel=9, max_depth=12,
                                              background=(0.5, 0.6, 0.8), vfov=50.0, lookfrom=(60, 120, 520), lookat=(0, 0, 0)))
    assert cam.image_height == 61
    scene = rt.Scene.from_graph(gpu_ctx, gp, wp, None)
    out = {}
    for pipe in (_capi.PIPELINE_FUSED, _capi.PIPELINE_WAVEFRONT):
        gpu_ctx.reset_stats()
        out[pipe] = (scene.render(cam, rt.render_opts(cam, seed=31, pipeline=pipe)), gpu_ctx.stats())
    (a, sa), (b, sb) = out[_capi.PIPELINE_FUSED], out[_capi.PIPELINE_WAVEFRONT]
    assert sa["paths"] == sb["paths"] == 97 * 61 * 9
    assert sa["segments"] == sb["segments"]
    np.testing.assert_allclose(b, a, rtol=1e-11, atol=1e-13)
    assert a.sum() > 0
    scene.close()


@pytest.mark.parametrize("wide,top", [(None, None), ("0", None), ("0", "0")], ids=["folded-table", "lds-top", "program-records"])
def test_head_section_in_global_memory_walks_trace_the_same_paths(gpu_ctx, monkeypatch, wide, top):
    """A head section too large for LDS (the stress scene's BVH) is walked from global memory over the table that
    folds two levels of its tree into one record (the default), or box by box over the program's own records with the
    first levels of the tree in an LDS table (RTX_WF_WIDE=0) or without it (RTX_WF_TOP=0 as well).  All of them, at
    several queue capacities, must trace the fused pipeline's paths, which walks the program in the reference's order."""
    gp = rt.Graph()
    wp, lp, cam = scenes.build("stress", gp, width=320, spp=4, depth=12, n_prims=200000)
    scene = rt.Scene.from_graph(gpu_ctx, gp, wp, lp)
    gpu_ctx.reset_stats()
    a = scene.render(cam, rt.render_opts(cam, seed=93, pipeline=_capi.PIPELINE_FUSED))
    sa = gpu_ctx.stats()
    if wide is not None:
        monkeypatch.setenv("RTX_WF_WIDE", wide)
    if top is not None:
        monkeypatch.setenv("RTX_WF_TOP", top)
    for cap in ("65536", "16384"):
        monkeypatch.setenv("RTX_WF_CAPACITY", cap)
        gpu_ctx.reset_stats()
        b = scene.render(cam, rt.render_opts(cam, seed=93, pipeline=_capi.PIPELINE_WAVEFRONT))
        sb = gpu_ctx.stats()
        assert sa["segments"] == sb["segments"] and sa["pdf_probes"] == sb["pdf_probes"] and sb["head_launches"] > 0
        np.testing.assert_allclose(b, a, rtol=1e-11, atol=1e-13)
    assert a.sum() > 0
    scene.close()


@pytest.mark.parametrize("name,kw,near", [("final", dict(width=240, spp=16, depth=30), "2"),
                                          ("stress", dict(width=320, spp=4, depth=12, n_prims=200000), "1")])
def test_rays_handed_back_by_the_nearer_child_first_walk_change_nothing(gpu_ctx, monkeypatch, name, kw, near):
    """The nearer-child-first walks walk a ray once more, in the reference's order, when its result might depend on
    the visiting order (a hit not clearly behind its box's entry: relative margin 1e-9).  With the margin widened to
    5 % (RTX_WF_NEAR_MARGIN, a testing knob) a large share of the rays takes that way, in the deferred section and
    (final scene: RTX_WF_NEAR=2; stress scene: by default, over the folded table of its BVH) in the head section; the
    paths must still be the fused pipeline's.  With the default margin random camera paths essentially never qualify."""
    gp = rt.Graph()
    wp, lp, cam = scenes.build(name, gp, **kw)
    scene = rt.Scene.from_graph(gpu_ctx, gp, wp, lp)
    gpu_ctx.reset_stats()
    a = scene.render(cam, rt.render_opts(cam, seed=91, pipeline=_capi.PIPELINE_FUSED))
    sa = gpu_ctx.stats()
    monkeypatch.setenv("RTX_WF_NEAR", near)
    monkeypatch.setenv("RTX_WF_STREAM", "1")
    for margin, lo, hi in (("0.05", 1000, None), (None, 0, 20)):
        if margin is not None:
            monkeypatch.setenv("RTX_WF_NEAR_MARGIN", margin)
        else:
            monkeypatch.delenv("RTX_WF_NEAR_MARGIN")
        gpu_ctx.reset_stats()
        b = scene.render(cam, rt.render_opts(cam, seed=91, pipeline=_capi.PIPELINE_WAVEFRONT))
        sb = gpu_ctx.stats()
        assert sa["segments"] == sb["segments"] and sa["pdf_probes"] == sb["pdf_probes"]
        assert sb["order_redos"] >= lo and (hi is None or sb["order_redos"] <= hi), (margin, sb["order_redos"])
        np.testing.assert_allclose(b, a, rtol=1e-11, atol=1e-13)
    scene.close()


def test_parallel_pipelines_trace_the_same_paths(gpu_ctx, monkeypatch):
    """A render call may be split by sample index between several wavefront pipelines that run side by side on
    streams of their own (RTX_WF_LANES; by default large calls use up to four).  Paths are addressed by (pixel,
    sample), so the split changes nothing but the order of the additions; the workspace grows when a later call
    needs more paths in flight."""
    gp = rt.Graph()
    wp, lp, cam = scenes.build("cornell", gp, width=64, spp=25, depth=12)
    scene = rt.Scene.from_graph(gpu_ctx, gp, wp, lp)
    out = {}
    for lanes, cap in (("1", "4096"), ("3", "1024"), ("4", "65536"), ("8", "2048")):
        monkeypatch.setenv("RTX_WF_LANES", lanes)
        monkeypatch.setenv("RTX_WF_CAPACITY", cap)
        gpu_ctx.reset_stats()
        out[lanes] = (scene.render(cam, rt.render_opts(cam, seed=21, pipeline=_capi.PIPELINE_WAVEFRONT)), gpu_ctx.stats())
    base, sbase = out["1"]
    assert sbase["paths"] == 64 * 64 * 25 and base.sum() > 0
    for lanes in ("3", "4", "8"):
        img, st = out[lanes]
        assert st["paths"] == sbase["paths"] and st["segments"] == sbase["segments"] and st["pdf_probes"] == sbase["pdf_probes"]
        np.testing.assert_allclose(img, base, rtol=1e-11, atol=1e-13)
    scene.close()


@pytest.mark.parametrize("name,kw", [("final", dict(width=160, spp=16, depth=40)), ("smoke", dict(width=128, spp=16, depth=30)),
                                     ("cornell", dict(width=128, spp=16, depth=30))])
def test_finishing_kernels_trace_the_same_paths(gpu_ctx, monkeypatch, name, kw):
    """Once a pipeline has no new camera paths left and few survivors, one launch runs them to their end: walking the
    program from LDS, the world program and the deferred section staged in turn (the default where both fit), or from
    global memory (RTX_WF_FINISH_LDS=0); it takes over at RTX_WF_FINISH survivors.  Early, late or never, with either
    kernel, the paths are the fused pipeline's (final: media of both kinds of boundary next to the deferred cluster;
    smoke: box-bounded media; cornell: no media)."""
    gp = rt.Graph()
    wp, lp, cam = scenes.build(name, gp, **kw)
    scene = rt.Scene.from_graph(gpu_ctx, gp, wp, lp)
    gpu_ctx.reset_stats()
    a = scene.render(cam, rt.render_opts(cam, seed=77, pipeline=_capi.PIPELINE_FUSED))
    sa = gpu_ctx.stats()
    monkeypatch.setenv("RTX_WF_LANES", "2")
    for lds, finish in (("1", "4194304"), ("1", "8192"), ("0", "4194304"), ("1", "256")):
        monkeypatch.setenv("RTX_WF_FINISH_LDS", lds)
        monkeypatch.setenv("RTX_WF_FINISH", finish)
        gpu_ctx.reset_stats()
        b = scene.render(cam, rt.render_opts(cam, seed=77, pipeline=_capi.PIPELINE_WAVEFRONT))
        sb = gpu_ctx.stats()
        assert sa["segments"] == sb["segments"] and sa["pdf_probes"] == sb["pdf_probes"], (lds, finish)
        np.testing.assert_allclose(b, a, rtol=1e-11, atol=1e-13)
        # a threshold above the whole call hands every path to the finishing kernel after the first iterations
        if finish == "4194304":
            assert sb["iterations"] <= 2 * 8
    assert a.sum() > 0
    scene.close()


def test_calls_on_different_streams_do_not_overlap_in_the_workspace(gpu_ctx):
    """Two render calls of one context, back to back on two different streams: they share the context's wavefront
    workspace and counters, so the library orders the second behind the first (an event on the first call's stream).
    Results must be those of the same two calls made one after the other with a host wait in between."""
    import hiprt
    gp = rt.Graph()
    wp, lp, cam = scenes.build("final", gp, width=160, spp=16, depth=30)
    scene = rt.Scene.from_graph(gpu_ctx, gp, wp, lp)
    n = cam.image_height * cam.image_width * 3
    o1, o2 = rt.render_opts(cam, seed=51), rt.render_opts(cam, sample_begin=2, sample_end=11, seed=52)
    ref = []
    for o in (o1, o2):
        acc = hiprt.DeviceArray(n)
        scene.render_device(cam, o, acc.ptr)
        ref.append(acc.numpy())
    s1, s2 = hiprt.Stream(), hiprt.Stream()
    out = [hiprt.DeviceArray(n), hiprt.DeviceArray(n)]
    for _ in range(3):
        for acc in out:
            acc.zero()
        hiprt.synchronize()
        scene.render_device(cam, o1, out[0].ptr, s1.handle.value)
        scene.render_device(cam, o2, out[1].ptr, s2.handle.value)
        for got, want in zip(out, ref):
            np.testing.assert_allclose(got.numpy(), want, rtol=1e-11, atol=1e-13)
    s1.close(); s2.close()
    scene.close()


def test_sample_ranges_and_rectangles_add_up(gpu_ctx):
    """Sharding property: disjoint sample ranges and pixel rectangles sum to the full render (RNG streams are keyed
    by (pixel, sample), so the shards trace exactly the same paths)."""
    gp = rt.Graph()
    wp, lp, cam = scenes.build("cornell", gp, width=40, spp=36, depth=10)
    scene = rt.Scene.from_graph(gpu_ctx, gp, wp, lp)
    full = scene.render(cam, rt.render_opts(cam, seed=5))
    parts = np.zeros_like(full)
    for (b, e) in [(0, 7), (7, 20), (20, 36)]:
        scene.render(cam, rt.render_opts(cam, sample_begin=b, sample_end=e, seed=5), parts)
    np.testing.assert_allclose(parts, full, rtol=1e-12, atol=1e-12)
    tiles = np.zeros_like(full)
    for rect in [(0, 0, 17, 40), (17, 0, 40, 9), (17, 9, 40, 40)]:
        scene.render(cam, rt.render_opts(cam, rect=rect, seed=5), tiles)
    np.testing.assert_allclose(tiles, full, rtol=1e-12, atol=1e-12)
    empty = scene.render(cam, rt.render_opts(cam, sample_begin=3, sample_end=3, seed=5))
    assert not empty.any()
    scene.close()


def test_resolve_bit_exact(gpu_ctx):
    """color_to_rgb_with_samples: NaN scrub, the requested-spp scale, gamma 2, clamp, truncation."""
    rng = np.random.default_rng(3)
    acc = rng.uniform(-1.0, 1200.0, size=(37, 53, 3))
    acc[0, 0] = (np.nan, 0.0, -0.0)
    acc[1, 1] = (np.inf, 1e-300, 999.9999)
    ref = np.empty(acc.shape, dtype=np.uint8)
    H.oracle_lib().orc_resolve_rgb8(H.dptr(acc), 53, 37, 1000, ref.ctypes.data_as(H.PU8))
    out = gpu_ctx.resolve_rgb8(acc, 1000)
    assert (out == ref).all()


def test_errors_are_reported_not_thrown(gpu_ctx):
    gp = rt.Graph()
    wp, lp, cam = scenes.build("cornell", gp, width=16, spp=4, depth=4)
    desc = rt.flatten(gp, wp, lp)
    desc.world_end = desc.n_nodes + 5
    with pytest.raises(rt.RtxError) as e:
        rt.Scene(gpu_ctx, desc)
    assert e.value.status == _capi.RTX_ERR_INVALID and "world program" in str(e.value)
    desc = rt.flatten(gp, wp, lp)
    scene = rt.Scene(gpu_ctx, desc)
    with pytest.raises(rt.RtxError):
        scene.render(cam, rt.render_opts(cam, sample_end=cam.sqrt_spp ** 2 + 1))
    with pytest.raises(rt.RtxError):
        scene.render(cam, rt.render_opts(cam, rect=(0, 0, 17, 16)))
    scene.close()

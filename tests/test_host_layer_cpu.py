"""CPU tier: host-side steps either side of the hot path - the texture ingest of ImageTexture::new (image.rs:15-43)
and the identity a checkpoint of the accumulator stores."""
import ctypes as C
import os

import numpy as np

import raytracing_rust_amd as rt
from raytracing_rust_amd import _capi, scenes
import helpers as H


def _write_ppm(path, rgb):
    h, w, _ = rgb.shape
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (w, h))
        f.write(np.ascontiguousarray(rgb, dtype=np.uint8).tobytes())


def _texel_probe(g, lib, tex, n=64):
    """Texture::value on a grid of (u, v): tells which file was loaded."""
    rng = np.random.default_rng(9)
    uvp = np.ascontiguousarray(np.concatenate([rng.uniform(0, 1, (n, 2)), np.zeros((n, 3))], axis=1))
    out = np.empty((n, 3))
    if lib is None:
        desc = rt.flatten(g, g.list([g.sphere((0, 0, 0), 1.0, g.lambertian(tex))]), None)
        # the material's texture index in the flattened tables
        t = desc.materials[desc.n_materials - 1].tex
        H.hostsim_lib().sim_texture_value(C.byref(desc), t, H.dptr(uvp), n, H.dptr(out))
    else:
        assert lib.orc_tex_value(g.session, tex, H.dptr(uvp), n, H.dptr(out)) == 0
    return out


def _solid(rgb, w=5, h=3):
    a = np.empty((h, w, 3), dtype=np.uint8)
    a[:] = rgb
    return a


def test_image_texture_search_order_and_cyan_fallback(tmp_path, monkeypatch, capfd):
    """image.rs:17-21: the name itself, then textures/<name>, then ../textures/<name>; image.rs:30-35: then
    $RTW_IMAGES/<name>; image.rs:37-43,61-63: otherwise an error line and a texture that is cyan everywhere.  Each
    candidate holds a different colour, and the candidates are removed one by one."""
    work = tmp_path / "run" / "cwd"
    (work / "textures").mkdir(parents=True)
    (tmp_path / "run" / "textures").mkdir()
    (tmp_path / "env").mkdir()
    files = [(work / "t.ppm", (255, 0, 0)), (work / "textures" / "t.ppm", (0, 255, 0)),
             (tmp_path / "run" / "textures" / "t.ppm", (0, 0, 255)), (tmp_path / "env" / "t.ppm", (255, 255, 0))]
    for path, colour in files:
        _write_ppm(path, _solid(colour))
    monkeypatch.chdir(work)
    monkeypatch.setenv("RTW_IMAGES", str(tmp_path / "env"))
    expected = [np.array(c) / 255.0 for _, c in files] + [np.array([0.0, 1.0, 1.0])]
    for step in range(5):
        for lib in (None, H.oracle_lib()):
            g = rt.Graph() if lib is None else H.oracle_graph()
            val = _texel_probe(g, lib, g.image_file("t.ppm"))
            assert np.abs(val - expected[step]).max() < 1e-15, (step, "product" if lib is None else "oracle", val[0])
        if step < 4:
            os.remove(files[step][0])
    assert "Could not load image file 't.ppm'" in capfd.readouterr().err
    # without the variable the fourth place is not searched
    _write_ppm(files[3][0], _solid(files[3][1]))
    monkeypatch.delenv("RTW_IMAGES")
    g = rt.Graph()
    assert np.abs(_texel_probe(g, None, g.image_file("t.ppm")) - expected[4]).max() < 1e-15


def test_image_texture_file_lookup_is_nearest_with_v_flipped(tmp_path, monkeypatch):
    """image.rs:66-74 through the file path: u -> column, 1 - v -> row, truncation, clamped to the last texel."""
    rng = np.random.default_rng(10)
    img = rng.integers(0, 256, size=(7, 11, 3), dtype=np.uint8)
    _write_ppm(tmp_path / "grid.ppm", img)
    monkeypatch.chdir(tmp_path)
    g = rt.Graph()
    tex = g.image_file("grid.ppm")
    desc = rt.flatten(g, g.list([g.sphere((0, 0, 0), 1.0, g.lambertian(tex))]), None)
    uv = np.array([[0.0, 0.0], [1.0, 1.0], [0.5, 0.5], [0.999, 0.001], [0.09, 0.86], [-3.0, 7.0]])
    uvp = np.ascontiguousarray(np.concatenate([uv, np.zeros((len(uv), 3))], axis=1))
    out = np.empty((len(uv), 3))
    H.hostsim_lib().sim_texture_value(C.byref(desc), desc.materials[desc.n_materials - 1].tex, H.dptr(uvp), len(uv), H.dptr(out))
    for (u, v), got in zip(uv, out):
        uc, vc = min(max(u, 0.0), 1.0), 1.0 - min(max(v, 0.0), 1.0)
        i, j = min(int(uc * 11), 10), min(int(vc * 7), 6)
        assert (got == img[j, i] * (1.0 / 255.0)).all(), (u, v)


def _fingerprint(g, world, lights, cam):
    desc = rt.flatten(g, world, lights)
    return _capi.lib.rtxh_checkpoint_fingerprint(C.byref(desc), C.byref(cam))


def test_checkpoint_identity_covers_every_table():
    """A checkpoint must not be resumed by a render of another scene: the stored identity changes with a medium's
    density, a transform's offset or angle, the light list, a texel, a Perlin table, and stays the same for the same
    scene built twice."""
    def build(density=0.01, offset=(265.0, 0.0, 295.0), angle=15.0, with_light=True, texel=7, noise_seed=3):
        g = rt.Graph()
        white = g.lambertian_color(0.73, 0.73, 0.73)
        img = np.full((2, 2, 3), texel, dtype=np.uint8)
        box = g.translate(g.rotate_y(g.box((0, 0, 0), (165, 330, 165), white), angle), offset)
        light = g.quad((113, 554, 127), (330, 0, 0), (0, 0, 305), g.diffuse_light_color(7, 7, 7))
        world = g.list([light, g.medium(box, density, g.solid(0, 0, 0)), g.sphere((100, 100, 100), 50.0, g.lambertian(g.image(img))),
                        g.sphere((300, 100, 100), 50.0, g.lambertian(g.noise(0.2, noise_seed)))])
        lights = g.list([g.quad((113, 554, 127), (330, 0, 0), (0, 0, 305), g.none())]) if with_light else None
        cam = rt.initialize_camera(rt.camera_desc(image_width=32, samples_per_pixel=4, max_depth=5, lookfrom=(278, 278, -800), lookat=(278, 278, 0), vfov=40.0))
        return _fingerprint(g, world, lights, cam)

    base = build()
    assert base == build() and base != 0
    changed = [build(density=0.02), build(offset=(265.0, 0.0, 296.0)), build(angle=16.0), build(with_light=False),
               build(texel=8), build(noise_seed=4)]
    assert all(c != base for c in changed) and len(set(changed)) == len(changed)

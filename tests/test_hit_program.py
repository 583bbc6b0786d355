"""CPU tier: the flattened hit program (host layer + device core compiled for the host) against the oracle's
recursive trait-object traversal.  Bit-exact first-hit AOVs; BVH shape; bounding boxes."""
import ctypes

import numpy as np
import pytest

import raytracing_rust_amd as rt
from raytracing_rust_amd import _capi, scenes
import helpers as H
import recipes


@pytest.mark.parametrize("name,width", [("cornell", 150), ("smoke", 150), ("final", 200)])
def test_named_scenes_bit_exact(name, width):
    go, gp = H.oracle_graph(), rt.Graph()
    wo, lo, cam_o = scenes.build(name, go, width=width, prefix="orc_", lib=H.oracle_lib())
    wp, lp, cam = scenes.build(name, gp, width=width)
    assert bytes(cam_o) == bytes(cam), "Camera::initialize differs between oracle and host layer"
    for time in (0.0, 0.73):
        rays = rt.pixel_center_rays(cam, time=time)
        H.assert_aov_equal(H.oracle_first_hit(go, wo, rays), H.hostsim_first_hit(rt.flatten(gp, wp, lp), rays), name)


def test_stress_scene_bit_exact():
    """The synthetic stress configuration at a reduced primitive count: the two scene builders draw the same scene
    (same camera, same counts), and the hit program finds the oracle's hits in a BVH of 30 000 mixed leaves."""
    go, gp = H.oracle_graph(), rt.Graph()
    wo, lo, cam_o = scenes.build("stress", go, width=256, n_prims=30000, prefix="orc_", lib=H.oracle_lib())
    wp, lp, cam = scenes.build("stress", gp, width=256, n_prims=30000)
    assert bytes(cam_o) == bytes(cam) and cam.image_height == 144
    d = rt.flatten(gp, wp, lp)
    assert d.n_spheres == 15000 and d.n_quads == 15000 and d.n_boxes == 32767 and d.n_lights == 64
    rays = rt.pixel_center_rays(cam, time=0.5)
    a = H.oracle_first_hit(go, wo, rays)
    H.assert_aov_equal(a, H.hostsim_first_hit(d, rays), "stress")
    assert np.isfinite(a["t"]).sum() > 2000


@pytest.mark.parametrize("recipe", recipes.ALL, ids=lambda r: r.__name__)
def test_recipes_bit_exact(recipe):
    go, gp = H.oracle_graph(), rt.Graph()
    wo, _, rays = recipe(go)
    wp, _, _ = recipe(gp)
    a = H.oracle_first_hit(go, wo, rays)
    b = H.hostsim_first_hit(rt.flatten(gp, wp, None), rays)
    H.assert_aov_equal(a, b, recipe.__name__)
    assert np.isfinite(a["t"]).sum() > 50, "recipe rays should hit something"


@pytest.mark.parametrize("name,width", [("final", 160), ("stress", 200)])
def test_staged_walk_with_streamed_sections_is_the_same_walk(name, width):
    """The wavefront pipeline walks the world program in stages: the head section (the world's first large box subtree)
    as a stream in which a lane takes the next ray as soon as its own has left the section, the rest of the program
    from the hit found so far, the deferred section as a stream over the rays that entered it.  First hits must be
    those of the undivided walk, bit for bit, and both kinds of section must actually occur."""
    gp = rt.Graph()
    wp, lp, cam = scenes.build(name, gp, width=width, n_prims=20000)
    d = rt.flatten(gp, wp, lp)
    rng = np.random.default_rng(12)
    rays = rt.pixel_center_rays(cam, time=0.4)
    # secondary-like rays: from random points of the scene's extent in random directions
    lo, hi = (np.array([-1000.0, 0.0, -1000.0]), np.array([1000.0, 600.0, 1000.0])) if name == "final" else (np.full(3, -1000.0), np.full(3, 1000.0))
    sec = np.concatenate([rng.uniform(lo, hi, (6000, 3)), rng.normal(size=(6000, 3)), rng.uniform(0, 1, (6000, 1))], axis=1)
    rays = np.concatenate([rays, sec])
    whole = H.hostsim_first_hit(d, rays)
    staged, streamed = H.hostsim_first_hit_staged(d, rays)
    H.assert_aov_equal(whole, staged, name)
    assert streamed >= len(rays)                                  # the head section saw every ray
    if name == "final":
        assert streamed > len(rays) + 100                         # and some rays entered the deferred section
    assert np.isfinite(whole["t"]).sum() > 1000


@pytest.mark.parametrize("name", ["final", "ties", "stress"])
def test_nearer_child_first_finds_the_same_hits(name):
    """Box trees of the head and deferred sections walked nearer child first (walk_stream_near) against the reference's
    fixed left-to-right order (bvh.rs:94-102): the same first hit, bit for bit, down to which of several primitives
    with exactly equal t is reported (the first sphere, the last quad: sphere.rs:99-101, quad.rs:74).  The recipe is
    built of such ties: coincident quads and spheres.  (Not covered, and not guaranteed: a tie in which the hit lies on
    the very boundary of its own bounding box, where the box test itself depends on the tmax the order has reached.)"""
    gp = rt.Graph()
    if name == "final":
        wp, lp, cam = scenes.build("final", gp, width=120)
        rng = np.random.default_rng(31)
        sec = np.concatenate([rng.uniform([-1000.0, 0.0, -1000.0], [1000.0, 600.0, 1000.0], (8000, 3)), rng.normal(size=(8000, 3)),
                              rng.uniform(0, 1, (8000, 1))], axis=1)
        rays = np.concatenate([rt.pixel_center_rays(cam, time=0.7), sec])
    elif name == "stress":   # one BVH over spheres and general quads: it is the head section, nothing is deferred
        wp, lp, cam = scenes.build("stress", gp, width=160, n_prims=30000)
        rng = np.random.default_rng(32)
        sec = np.concatenate([rng.uniform(-1000.0, 1000.0, (8000, 3)), rng.normal(size=(8000, 3)), rng.uniform(0, 1, (8000, 1))], axis=1)
        rays = np.concatenate([rt.pixel_center_rays(cam, time=0.2), sec])
    else:
        wp, lp, rays = recipes.ties_in_box_trees(gp)
    d = rt.flatten(gp, wp, lp)
    whole = H.hostsim_first_hit(d, rays)
    lib = H.hostsim_lib()
    # the stress scene's BVH does not fit LDS; its boxes are then read from a table: two levels of the tree folded into
    # one record (table 2, the default), or the first levels in heap order (1), or from the program (0).  Same hits.
    for near, walks, table in ((1, 0 if name == "stress" else 1, 2), (2, 1 if name == "stress" else 2, 2), (2, 1 if name == "stress" else 2, 1),
                               (2, 1 if name == "stress" else 2, 0)):
        tops, wides = lib.sim_top_walks(), lib.sim_wide_walks()
        lib.sim_set_head_table(table)
        try:
            staged, streamed = H.hostsim_first_hit_staged(d, rays, near=near)
        finally:
            lib.sim_set_head_table(2)
        assert lib.sim_near_walks() == walks
        through = name == "stress" and near == 2
        assert lib.sim_wide_walks() - wides == (1 if through and table == 2 else 0)
        assert lib.sim_top_walks() - tops == (1 if through and table == 1 else 0)
        H.assert_aov_equal(whole, staged, f"{name} near={near} table={table}")
        assert streamed >= len(rays)
    hit = np.isfinite(whole["t"])
    assert hit.sum() > 1500
    if name == "ties":   # both copies of the doubled quads and spheres do win somewhere?  No: always the same copy
        ids = set(np.unique(whole["prim_id"][hit]).tolist())
        quads = [6 * 64 + 2 * k for k in range(16)]              # creation ordinals: 64 boxes of six faces, then the quad pairs
        assert sum((q in ids) + (q + 1 in ids) for q in quads) == sum((q in ids) or (q + 1 in ids) for q in quads) >= 12


def test_folded_table_walk_hands_rays_back_to_the_program_order():
    """The walk over the folded table (walk_stream_wide) re-walks a ray it cannot vouch for over the program itself,
    in the reference's order.  With the watch's margin opened wide most rays whose hit lies near the entry of its
    leaf box go that way; the hits must not change, bit for bit."""
    gp = rt.Graph()
    wp, lp, cam = scenes.build("stress", gp, width=160, n_prims=30000)
    rng = np.random.default_rng(33)
    sec = np.concatenate([rng.uniform(-1000.0, 1000.0, (8000, 3)), rng.normal(size=(8000, 3)), rng.uniform(0, 1, (8000, 1))], axis=1)
    rays = np.concatenate([rt.pixel_center_rays(cam, time=0.2), sec])
    rays[5, 3] = np.inf                                     # a ray that is not finite goes the program's way at once
    # rays from far outside are served by the table all the same (its tests measure from the root box's entry), up to
    # 1e9 times the section's extent; beyond that they go the program's way too
    for k, o in enumerate([(3e8, 1e8, -2e8), (-4e13, 10.0, 20.0), (0.0, 6e8, 0.0), (2e6, -3e6, 1e6)]):
        rays[6 + k, 0:3] = o
        rays[6 + k, 3:6] = np.array([30.0 * k, -20.0, 10.0]) - np.array(o)
    # rays parallel to an axis (a component of the direction below 1e-8 in magnitude: Aabb::hit skips that axis,
    # aabb.rs:142-144) go the program's way as well: the table's box test has no case for skipped axes
    d = rt.flatten(gp, wp, lp)
    for k, dvec in enumerate([(1.0, 0.0, 0.0), (0.0, -1.0, 0.0), (0.0, 0.0, 1.0), (0.3, 1e-9, -0.8)]):
        centre = np.array(d.spheres[7 * k].c0[:])             # aimed at a sphere's centre: they hit something
        rays[10 + k, 0:3] = centre - 3000.0 * np.array(dvec)
        rays[10 + k, 3:6] = dvec
        rays[10 + k, 6] = 0.0
    whole = H.hostsim_first_hit(d, rays)
    assert np.isfinite(whole["t"][10:14]).all()
    lib = H.hostsim_lib()
    lib.sim_set_near_margin.argtypes = [ctypes.c_double]
    seen = []
    for margin in (0.0, 0.05, 1.0):
        lib.sim_set_near_margin(margin)
        try:
            wides = lib.sim_wide_walks()
            staged, _ = H.hostsim_first_hit_staged(d, rays, near=2)
        finally:
            lib.sim_set_near_margin(0.0)
        assert lib.sim_wide_walks() - wides == 1
        H.assert_aov_equal(whole, staged, f"folded table, margin {margin}")
        seen.append(lib.sim_order_redos())
    assert 1 <= seen[0] <= 10 and seen[1] > 50 and seen[2] > seen[1], seen   # margin 0: the odd rays above, those that pass the root box


@pytest.mark.parametrize("n_prims", [30000, 30001, 50021])
def test_folded_table_covers_the_program(n_prims):
    """The table that folds two levels of a large head section's box tree into one record, against the program it was
    made from: every record reachable once, depth first; every slot's float box strictly around the program box it
    stands for; the boxes left out are exactly those whose two children are boxes; every leaf of the tree appears
    once, in the program's order, with the position and end of its units; the stated depth is the deepest record's."""
    gp = rt.Graph()
    wp, lp, cam = scenes.build("stress", gp, width=64, n_prims=n_prims)
    d = rt.flatten(gp, wp, lp)
    lib = H.hostsim_lib()
    lib.sim_check_wide_table.restype = ctypes.c_long
    nrec, depth = ctypes.c_uint32(0), ctypes.c_uint32(0)
    leaves = lib.sim_check_wide_table(ctypes.byref(d), ctypes.byref(nrec), ctypes.byref(depth))
    assert leaves > 0, leaves
    # a BvhNode over n objects has one lowest box per pair of objects (bvh.rs:36-57); boxes with two box children get a record
    assert leaves == 2 ** 14 if n_prims <= 2 ** 15 else leaves == 2 ** 15
    assert 3 <= depth.value <= 10 and leaves // 4 < nrec.value < leaves
    # the shipped scenes are walked from LDS and get no such table
    gq = rt.Graph()
    wq, lq, _ = scenes.build("final", gq, width=32)
    assert lib.sim_check_wide_table(ctypes.byref(rt.flatten(gq, wq, lq)), None, None) == 0


def test_nearer_child_first_hands_back_what_depends_on_the_order():
    """The one case in which the visiting order can change a closest hit: a hit on the boundary of its own bounding box
    that ties with another (recipes.ties_on_box_boundaries is made of them).  The nearer-child-first walk notices such
    rays (a candidate hit not behind its box's entry, or a box culled by a hair) and hands them back; they are walked
    again in the reference's order, so the staged walk still equals the undivided fixed-order walk and the oracle, bit
    for bit.  (With the watch compiled out, 79 of these rays report the other sphere of their pair.)"""
    go, gp = H.oracle_graph(), rt.Graph()
    wo, _, rays = recipes.ties_on_box_boundaries(go)
    wp, lp, _ = recipes.ties_on_box_boundaries(gp)
    d = rt.flatten(gp, wp, lp)
    whole = H.hostsim_first_hit(d, rays)
    H.assert_aov_equal(H.oracle_first_hit(go, wo, rays), whole, "boundary ties, fixed order")
    staged, streamed = H.hostsim_first_hit_staged(d, rays, near=1)
    assert H.hostsim_lib().sim_near_walks() == 1
    redos = H.hostsim_lib().sim_order_redos()
    H.assert_aov_equal(whole, staged, "boundary ties, nearer child first")
    assert 300 <= redos < 1000, redos                     # the axis rays of the 100 cells, not the random ones
    tie = np.isclose(whole["t"][:500:5], 4.0) & (rays[:500:5, 3] == 1.0)
    assert tie.sum() == 100                                # the pairs do tie where the recipe says


def test_final_scene_bvh_shape():
    """400 leaves -> 511 nodes, 1000 leaves -> 1023 nodes (f(1)=f(2)=1, f(n)=1+f(n/2)+f(n-n/2)), bvh.rs:26-71."""
    g = rt.Graph()
    w, l, cam = scenes.build("final", g, width=32)
    d = rt.flatten(g, w, l)
    assert d.n_boxes == 511 + 1023
    assert d.n_spheres == 1007 and d.n_quads == 2402
    assert d.n_media == 2 and d.n_lights == 1 and d.has_lights == 1
    kinds = np.array([d.nodes[i].kind_idx >> 28 for i in range(d.n_nodes)])
    assert (kinds[: d.world_end] == _capi.NODE_MEDIUM).sum() == 2
    assert (kinds == _capi.NODE_ENTER).sum() == 2 and (kinds == _capi.NODE_EXIT).sum() == 2   # rotate inside translate
    for i in np.nonzero(kinds == _capi.NODE_BOX)[0]:
        assert i < d.nodes[i].skip <= d.n_nodes


def test_bounding_boxes_match_oracle():
    """Hittable::bounding_box of every implementor, including padding of flat quads and the rotated-corner box."""
    def recipe(g):
        m = g.lambertian_color(0.5, 0.5, 0.5)
        q = g.quad((1, 2, 3), (4, 0, 0), (0, 0, 5), m)                       # flat in y: padded by 1e-4
        s = g.sphere_moving((0, 0, 0), (3, 1, 0), 2.0, m)
        b = g.box((0, 0, 0), (1, 2, 3), m)
        r = g.rotate_y(b, 30.0)
        t = g.translate(r, (10, 20, 30))
        lst = g.list([q, s, t])
        return [q, s, b, r, t, lst, g.bvh(lst), g.medium(s, 0.1, g.solid(1, 1, 1)), g.list(), g.translate(g.list(), (1, 1, 1))]
    go, gp = H.oracle_graph(), rt.Graph()
    for ho, hp in zip(recipe(go), recipe(gp)):
        bo, bp = go.bounding_box(ho), gp.bounding_box(hp)
        assert (bo is None) == (bp is None)
        if bo is not None:
            assert (bo.view(np.uint64) == bp.view(np.uint64)).all()


def test_unsupported_graphs_are_rejected():
    g = rt.Graph()
    m = g.lambertian_color(0.5, 0.5, 0.5)
    inner = g.medium(g.sphere((0, 0, 0), 1.0, m), 0.5, g.solid(1, 1, 1))
    outer = g.medium(g.list([g.sphere((0, 0, 0), 5.0, m), inner]), 0.5, g.solid(1, 1, 1))
    with pytest.raises(rt.RtxError) as e:
        rt.flatten(g, g.list([outer]), None)
    assert e.value.status == _capi.RTX_ERR_UNSUPPORTED
    with pytest.raises(rt.GraphError):
        g.bvh(g.list())                     # the reference would index an empty slice
    with pytest.raises(rt.GraphError):
        g.sphere((0, 0, 0), 1.0, 9999)      # bad material handle


@pytest.mark.parametrize("name", ["final", "cornell"])
def test_flat_stretch_swept_in_step_is_the_same_walk(name):
    """The stretch of the world program behind its head section is a list every ray passes unit by unit; the
    wavefront pipeline's world-program kernel sweeps it with one position for the whole wave (walk_flat).  Stage by
    stage (head section, flat stretch, deferred section) against the undivided walk: same hit bit for bit, media
    included, on camera rays and on rays from inside the scene (which meet the media and the sphere cluster)."""
    g = rt.Graph()
    w, l, cam = scenes.build(name, g, width=96)
    desc = rt.flatten(g, w, l)
    rng = np.random.default_rng(17)
    inside = np.concatenate([rng.uniform(-100, 600, (6000, 3)), rng.normal(size=(6000, 3)) * rng.uniform(0.1, 5.0, (6000, 1)),
                             rng.uniform(0, 1, (6000, 1))], axis=1)
    odd = np.array([[278.0, 278.0, -800.0, 0.0, 0.0, 1.0, 0.5], [0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0], [278.0, 278.0, -800.0, np.nan, 0.0, 1.0, 0.5],
                    [278.0, 278.0, -800.0, np.inf, 0.0, 1.0, 0.5], [1e12, 0.0, 0.0, -1.0, 0.0, 0.0, 0.2]])
    rays = np.ascontiguousarray(np.concatenate([rt.pixel_center_rays(cam, time=0.4), inside, odd]))
    lib = H.hostsim_lib()
    lib.sim_check_flat_walk.restype = ctypes.c_long
    lib.sim_check_flat_walk.argtypes = [ctypes.POINTER(_capi.SceneDesc), H.PD, ctypes.c_size_t, ctypes.c_uint32, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]
    hits, media = ctypes.c_size_t(0), ctypes.c_size_t(0)
    for bounce in (0, 4):
        bad = lib.sim_check_flat_walk(ctypes.byref(desc), H.dptr(rays), len(rays), bounce, ctypes.byref(hits), ctypes.byref(media))
        assert bad == 0, f"{bad} of {len(rays)} rays differ"
        assert hits.value > len(rays) // 2
        if name == "final":
            assert media.value > 200


def test_head_section_with_transformed_leaves_is_not_swept():
    """The sweep's kernels walk the head section without transform contexts or media; a head section that holds any
    (a BVH whose leaves are translated and rotated) must leave the program to the general walk."""
    g = rt.Graph()
    w, _, _ = recipes.nested_transforms(g)
    desc = rt.flatten(g, w, None)
    lib = H.hostsim_lib()
    lib.sim_check_flat_walk.restype = ctypes.c_long
    lib.sim_check_flat_walk.argtypes = [ctypes.POINTER(_capi.SceneDesc), H.PD, ctypes.c_size_t, ctypes.c_uint32, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]
    rays = np.zeros((1, 7))
    assert lib.sim_check_flat_walk(ctypes.byref(desc), H.dptr(rays), 1, 0, None, None) == -1

"""Deterministic scene recipes for the parity tests.  Each takes a Graph (oracle- or product-backed), builds the same
graph through the reference-shaped constructors and returns (world, lights, rays[n,7])."""
import numpy as np


def _rays_towards(rng, n, center, radius, dist):
    """Random rays from a shell of radius `dist` aimed near `center` (unnormalised directions, random time)."""
    o = rng.normal(size=(n, 3))
    o = center + dist * o / np.linalg.norm(o, axis=1, keepdims=True)
    target = center + rng.uniform(-radius, radius, size=(n, 3))
    d = (target - o) * rng.uniform(0.2, 3.0, size=(n, 1))
    return np.concatenate([o, d, rng.uniform(0, 1, size=(n, 1))], axis=1)


def _materials(g):
    return [g.lambertian_color(0.5, 0.5, 0.5), g.metal(0.8, 0.8, 0.9, 0.3), g.dielectric(1.5),
            g.diffuse_light_color(4, 4, 4), g.none()]


def _random_leaf(g, rng, mats, span=100.0):
    m = mats[int(rng.integers(len(mats)))]
    kind = int(rng.integers(4))
    c = rng.uniform(-span, span, size=3)
    if kind == 0:
        return g.sphere(c, float(rng.uniform(2, 15)), m)
    if kind == 1:
        return g.sphere_moving(c, c + rng.uniform(-10, 10, size=3), float(rng.uniform(2, 15)), m)
    if kind == 2:
        return g.quad(c, rng.uniform(-30, 30, size=3), rng.uniform(-30, 30, size=3), m)
    return g.box(c, c + rng.uniform(-25, 25, size=3), m)


def flat_random_prims(g):
    rng = np.random.default_rng(101)
    mats = _materials(g)
    world = g.list([_random_leaf(g, rng, mats) for _ in range(60)])
    return world, None, _rays_towards(rng, 6000, np.zeros(3), 110.0, 400.0)


def bvh_mixed_leaves(g):
    """BVH over 37 mixed leaves: exercises spans 1, 2 and 3 of the median split and list leaves (boxes)."""
    rng = np.random.default_rng(102)
    mats = _materials(g)
    leaves = g.list([_random_leaf(g, rng, mats) for _ in range(37)])
    world = g.list([g.bvh(leaves), g.sphere((0, 0, 0), 20.0, mats[2])])
    return world, None, _rays_towards(rng, 6000, np.zeros(3), 110.0, 350.0)


def nested_transforms(g):
    """Translate(RotateY(Translate(..))), a BVH whose leaves are transformed BVHs, and a transformed single-leaf BVH."""
    rng = np.random.default_rng(103)
    mats = _materials(g)
    inner = g.bvh(g.list([_random_leaf(g, rng, mats, 40.0) for _ in range(9)]))
    a = g.translate(g.rotate_y(g.translate(inner, (5, -3, 7)), 33.0), (60, 10, -20))
    b = g.rotate_y(g.translate(g.rotate_y(g.box((0, 0, 0), (30, 50, 20), mats[0]), -71.0), (-80, 0, 30)), 12.5)
    c = g.translate(g.bvh(g.list([g.sphere((0, 0, 0), 12.0, mats[1])])), (0, 70, 0))
    leaves = g.list([a, b, c] + [_random_leaf(g, rng, mats) for _ in range(6)])
    world = g.list([g.bvh(leaves), g.quad((-150, -60, -150), (300, 0, 0), (0, 0, 300), mats[0])])
    return world, None, _rays_towards(rng, 6000, np.zeros(3), 120.0, 420.0)


def ties_edges_and_axis_parallel(g):
    """Rays through box edges and corners, along axes (|d_k| < 1e-8 skips that slab axis), starting on surfaces,
    plus degenerate rays (zero and NaN directions)."""
    rng = np.random.default_rng(104)
    mats = _materials(g)
    boxes = [g.box((x, 0, z), (x + 10, 10 + ((x * 7 + z * 3) % 5), z + 10), mats[0]) for x in range(-30, 30, 10) for z in range(-30, 30, 10)]
    world = g.list([g.bvh(g.list(boxes)), g.sphere((0, 25, 0), 10.0, mats[2]),
                    g.quad((-30, 0, -30), (60, 0, 0), (0, 0, 60), mats[0])])
    rays = []
    for x in range(-30, 31, 10):          # straight down onto shared box edges / corners (two exact-zero components)
        for z in range(-30, 31, 5):
            rays.append([x, 50, z, 0, -1, 0, 0.0])
            rays.append([x, 50, z, 0, -2.5, 0, 0.5])
    for y in (0.0, 5.0, 10.0, 12.0):       # grazing along x and z at face heights
        for z in range(-30, 31, 10):
            rays.append([-100, y, z, 1, 0, 0, 0.0])
            rays.append([z, y, -100, 0, 0, 3, 0.0])
    for _ in range(500):                    # one tiny component (below and above the 1e-8 threshold)
        o = rng.uniform(-40, 40, size=3); o[1] = 60
        d = np.array([rng.uniform(-1, 1), -1.0, rng.choice([1e-9, -1e-9, 1e-7, 0.0, -0.0])])
        rays.append([*o, *d, 0.0])
    for _ in range(500):                    # origins exactly on the ground quad / sphere surface
        p = np.array([rng.uniform(-30, 30), 0.0, rng.uniform(-30, 30)])
        rays.append([*p, *rng.normal(size=3), 0.0])
    rays.append([0, 50, 0, 0, 0, 0, 0.0])
    rays.append([0, 50, 0, np.nan, -1, 0, 0.0])
    rays.append([np.nan, 50, 0, 0, -1, 0, 0.0])
    rays.append([0, 50, 0, np.inf, -1, 0, 0.0])
    return world, None, np.array(rays, dtype=np.float64)


def empty_lists_and_media(g):
    """Empty lists at several places, a medium (skipped by the deterministic AOV) and its boundary shared with the
    world, as in the reference's final scene."""
    rng = np.random.default_rng(105)
    mats = _materials(g)
    boundary = g.sphere((0, 0, 0), 30.0, mats[2])
    fog = g.medium(g.sphere((0, 0, 0), 500.0, mats[2]), 0.001, g.solid(1, 1, 1))
    smoke_box = g.medium(g.translate(g.rotate_y(g.box((0, 0, 0), (20, 20, 20), mats[0]), 20.0), (40, 0, 0)), 0.05, g.solid(0, 0, 0))
    inner = g.list([g.list(), boundary, g.medium(boundary, 0.2, g.solid(0.2, 0.4, 0.9)), g.list()])
    world = g.list([g.list(), inner, fog, smoke_box, g.bvh(g.list([g.list([g.sphere((60, 0, 60), 10.0, mats[0])])]))])
    return world, None, _rays_towards(rng, 3000, np.zeros(3), 70.0, 200.0)


def large_bvh_beyond_lds(g):
    """Two BVHs over 3000 spheres and 1500 boxes: the compiled program (about 0.7 MB) does not fit the 160 KB of LDS, so
    the traversal kernels read it from global memory; the transformed box BVH comes last and becomes the deferred
    section."""
    rng = np.random.default_rng(106)
    mats = _materials(g)
    spheres = g.list([g.sphere(rng.uniform(-200, 200, size=3), float(rng.uniform(2, 9)), mats[int(rng.integers(3))])
                      for _ in range(3000)])
    boxes = []
    for _ in range(1500):
        c = rng.uniform(-150, 150, size=3)
        boxes.append(g.box(c, c + rng.uniform(3, 20, size=3), mats[0]))
    world = g.list([g.bvh(spheres), g.quad((-300, -250, -300), (600, 0, 0), (0, 0, 600), mats[3]),
                    g.translate(g.rotate_y(g.bvh(g.list(boxes)), 25.0), (30, 0, -40))])
    return world, None, _rays_towards(rng, 8000, np.zeros(3), 220.0, 700.0)


ALL = [flat_random_prims, bvh_mixed_leaves, nested_transforms, ties_edges_and_axis_parallel, empty_lists_and_media,
       large_bvh_beyond_lds]


def ties_in_box_trees(g):
    """The shapes of the final scene's head and deferred sections with many hits of exactly equal t: a ground of 8 x 8
    boxes and 16 floating axis-aligned quads, each quad there twice (different materials), in one BVH; a light; a
    translated BVH of 64 overlapping spheres, half of them there twice.  Which of two coincident primitives a ray
    reports is decided by the order of the walk (the first sphere, the last quad)."""
    rng = np.random.default_rng(107)
    mats = _materials(g)
    leaves = []
    for i in range(8):
        for j in range(8):
            h = float(rng.uniform(5.0, 25.0))
            leaves.append(g.box((-80.0 + 20.0 * i, 0.0, -80.0 + 20.0 * j), (-61.0 + 20.0 * i, h, -61.0 + 20.0 * j), mats[0]))
    for i in range(4):
        for j in range(4):
            q = (-70.0 + 40.0 * i, 40.0 + 3.0 * ((i + j) % 3), -70.0 + 40.0 * j)
            leaves.append(g.quad(q, (25.0, 0, 0), (0, 0, 25.0), mats[0]))
            leaves.append(g.quad(q, (25.0, 0, 0), (0, 0, 25.0), mats[1]))
    order = rng.permutation(len(leaves))
    ground = g.bvh(g.list([leaves[k] for k in order]))
    centres = [tuple(c) for c in rng.uniform(0.0, 60.0, size=(64, 3))]
    centres += centres[:32]
    spheres = [g.sphere(c, 9.0, mats[k % 3]) for k, c in enumerate(centres)]
    cluster = g.translate(g.bvh(g.list(spheres)), (-30.0, 60.0, -30.0))
    world = g.list([ground, g.quad((-20, 200, -20), (40, 0, 0), (0, 0, 40), mats[3]), cluster])
    at_quads = _rays_towards(rng, 3000, np.array([0.0, 43.0, 0.0]), 80.0, 300.0)
    at_cluster = _rays_towards(rng, 3000, np.array([0.0, 90.0, 0.0]), 30.0, 250.0)
    inside = np.concatenate([rng.uniform([-30.0, 60.0, -30.0], [30.0, 120.0, 30.0], (2000, 3)), rng.normal(size=(2000, 3)), rng.uniform(0, 1, (2000, 1))], axis=1)
    return world, None, np.concatenate([at_quads, at_cluster, inside, _rays_towards(rng, 3000, np.array([0.0, 50.0, 0.0]), 90.0, 300.0)])


def texture_probe(g):
    """Every texture kind as the emission of a DiffuseLight (diffuse_light.rs:29-40: emitted = texture value at (u, v,
    p), on both faces): six quads facing the camera and two spheres.  A path of depth 1 returns exactly the texture
    value at its camera ray's hit, so a render with max_depth 1 probes the device's texture code pixel by pixel.
    Returns (world, camera keywords, texture handle per primitive in construction order)."""
    rng = np.random.default_rng(108)
    small = rng.integers(0, 256, size=(23, 37, 3), dtype=np.uint8)
    from raytracing_rust_amd import scenes
    checker = g.checker(0.32, g.solid(0.2, 0.3, 0.1), g.solid(0.9, 0.9, 0.9))          # checker.rs:25-33
    nested = g.checker(1.7, g.checker(0.21, g.solid(1.0, 0.1, 0.1), g.noise(4.0, 77)), g.image(small))
    marble = g.noise(0.2, 0x5EED0001)                                                    # final_scene.rs:141
    fine = g.noise(3.0, 5)
    earth = g.image(scenes.earth_pixels())                                               # final_scene.rs:132
    missing = g.image(None)                                                              # image.rs:61-63: cyan
    texs = [checker, nested, marble, fine, earth, missing, earth, nested]
    prims = []
    for k in range(6):
        x0, y0 = -7.5 + 5.0 * (k % 3), -5.0 + 5.0 * (k // 3)
        # not axis-aligned, one seen from behind: (u, v) = (alpha, beta) of quad.rs:78-86
        q = (x0, y0, 0.3 * k)
        u, v = ((4.5, 0.2, 0.4), (-0.1, 4.5, 0.3)) if k != 4 else ((0.0, 4.5, 0.0), (4.5, 0.0, 0.5))
        prims.append(g.quad(q, u, v, g.diffuse_light(texs[k])))
    prims.append(g.sphere((-4.0, 7.5, 1.0), 1.9, g.diffuse_light(texs[6])))             # sphere uv: sphere.rs:57-66
    prims.append(g.sphere_moving((4.0, 7.5, 1.0), (4.5, 7.9, 1.0), 1.9, g.diffuse_light(texs[7])))
    cam = dict(aspect_ratio=1.25, image_width=160, samples_per_pixel=4, max_depth=1, background=(0.1, 0.2, 0.3),
               vfov=60.0, lookfrom=(0.3, 1.5, 14.0), lookat=(0.0, 1.0, 0.0))
    return g.list(prims), cam, texs


def ties_on_box_boundaries(g):
    """Hits that lie ON the boundary of their own bounding box and tie with another hit: pairs of spheres that touch at a
    point where both touch their boxes (centres (0, y, z) radius 1 and (1, y, z) radius 2 meet at x = -1), hit by rays
    along the x axis, so that the box entry, and both hit parameters, are the same number to the last bit.  Which sphere
    the reference reports depends on the upper bound its left-to-right walk has reached when it tests the second box
    (bvh.rs:94-102: `t <= entry` culls it); small off-axis spheres vary the shape of the tree from cell to cell, so that
    a nearer-child-first walk visits some pairs in the other order.  The spheres form the last top-level BVH (the
    deferred section of the wavefront pipeline)."""
    rng = np.random.default_rng(109)
    mats = _materials(g)
    spheres, rays = [], []
    for i in range(10):
        for j in range(10):
            y, z = 10.0 * i, 10.0 * j
            cell = [g.sphere((0.0, y, z), 1.0, mats[0]), g.sphere((1.0, y, z), 2.0, mats[1])]
            for _ in range(int(rng.integers(0, 3))):                       # decoys the axis rays miss
                cell.append(g.sphere((float(rng.integers(-4, 6)), y + float(rng.choice([-3.0, 3.0])), z + float(rng.integers(-1, 2))), 0.5, mats[2]))
            for k in rng.permutation(len(cell)):
                spheres.append(cell[int(k)])
            for ox, dx in ((-5.0, 1.0), (-9.0, 2.0), (-3.0, 0.5), (-5.0, 4.0)):
                rays.append([ox, y, z, dx, 0.0, 0.0, 0.0])
            rays.append([8.0, y, z, -1.0, 0.0, 0.0, 0.0])                   # from the other side: no tie
    order = rng.permutation(len(spheres))
    world = g.list([g.quad((-50, -60, -50), (200, 0, 0), (0, 0, 200), mats[3]), g.bvh(g.list([spheres[int(k)] for k in order]))])
    rays = np.array(rays, dtype=np.float64)
    return world, None, np.concatenate([rays, _rays_towards(rng, 3000, np.array([0.0, 45.0, 45.0]), 60.0, 200.0)])


def material_probe(g):
    """Every material under a sphere light and a quad light, with a medium: metal with and without fuzz, glass (hit from
    outside and, by rays that start inside it, from inside at angles beyond the critical one), Lambertian, a
    box-bounded medium, a dense and a thin sphere-bounded medium, a moving sphere."""
    red, white = g.lambertian_color(0.65, 0.05, 0.05), g.lambertian_color(0.73, 0.73, 0.73)
    light = g.diffuse_light_color(9, 8, 7)
    floor = g.quad((-60, 0, -60), (120, 0, 0), (0, 0, 120), white)
    back = g.quad((-60, 0, -60), (120, 0, 0), (0, 80, 0), red)
    metal = g.sphere((-25, 10, 0), 10.0, g.metal(0.8, 0.85, 0.88, 0.0))
    fuzzy = g.sphere((25, 10, 0), 10.0, g.metal(0.7, 0.6, 0.5, 0.6))
    glass = g.sphere((0, 12, 10), 12.0, g.dielectric(1.5))
    mover = g.sphere_moving((0, 30, -20), (0, 36, -20), 5.0, white)
    smoke = g.medium(g.translate(g.rotate_y(g.box((0, 0, 0), (16, 16, 16), white), 25.0), (-20, 0, 25)), 0.08, g.solid(0.9, 0.9, 0.9))
    haze = g.medium(g.sphere((20, 8, 28), 8.0, white), 0.15, g.solid(0.2, 0.4, 0.9))
    # thin and large (mean free path 1000 against a diameter of 400): the device core draws its free path before it
    # intersects the boundary and mostly skips the intersection
    mist = g.medium(g.sphere((0, 20, 0), 200.0, white), 0.001, g.solid(0.8, 0.8, 0.8))
    quad_light = g.quad((-15, 70, -15), (30, 0, 0), (0, 0, 30), light)
    sphere_light = g.sphere((45, 45, 20), 6.0, light)
    world = g.list([floor, back, metal, fuzzy, glass, mover, smoke, haze, mist, quad_light, sphere_light])
    lights = g.list([g.quad((-15, 70, -15), (30, 0, 0), (0, 0, 30), g.none()), g.sphere((45, 45, 20), 6.0, g.none())])
    cam = dict(aspect_ratio=1.0, image_width=64, samples_per_pixel=16, max_depth=12, background=(0.02, 0.03, 0.05),
               vfov=55.0, lookfrom=(0.0, 35.0, 110.0), lookat=(0.0, 15.0, 0.0))
    return world, lights, cam

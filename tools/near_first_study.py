#!/usr/bin/env python3
"""Study (CPU, numpy): how many boxes and spheres a ray visits in the final scene's 1000-sphere cluster when the BVH of
bvh.rs:26-71 is walked in the reference's fixed left-to-right order, and when the nearer child is walked first.
Not part of the product: the shipped walk keeps the reference's order.  usage: python tools/near_first_study.py"""
import numpy as np, sys
rng = np.random.default_rng(1)
# the sphere cluster of final_scene.rs:149-172: 1000 spheres r=10, centres U[0,165)^3 (untransformed frame)
C = rng.uniform(0, 165, size=(1000, 3)); R = 10.0
lo = C - R; hi = C + R
class N: pass
def build(idx):
    n = N(); n.lo = lo[idx].min(0); n.hi = hi[idx].max(0)
    if len(idx) <= 2:
        n.leaf = list(idx) if len(idx) == 2 else [idx[0], idx[0]]; n.kids = None; return n
    s = n.hi - n.lo
    ax = 0 if (s[0] >= s[1] and s[0] >= s[2]) else (1 if s[1] >= s[2] else 2)
    srt = idx[np.argsort(lo[idx, ax], kind="stable")]; mid = len(idx)//2
    n.leaf = None; n.kids = (build(srt[:mid]), build(srt[mid:])); return n
root = build(np.arange(1000))
def slab(n, o, inv, tmin, tmax):
    t0 = (n.lo - o) * inv; t1 = (n.hi - o) * inv
    a = np.minimum(t0, t1); b = np.maximum(t0, t1)
    lo_ = max(tmin, a.max()); hi_ = min(tmax, b.min())
    return hi_ > lo_, a.max()
def sph(i, o, d, tmin, tmax):
    oc = o - C[i]; a = d @ d; hb = oc @ d; c = oc @ oc - R*R; disc = hb*hb - a*c
    if disc < 0: return None
    sq = np.sqrt(disc)
    for r in ((-hb - sq)/a, (-hb + sq)/a):
        if tmin < r < tmax: return r
    return None
def walk(n, o, d, inv, st, near):
    st["box"] += 1
    ok, _ = slab(n, o, inv, 0.001, st["t"])
    if not ok: return
    if n.kids is None:
        for i in n.leaf:
            st["sph"] += 1
            t = sph(i, o, d, 0.001, st["t"])
            if t is not None: st["t"] = t
        return
    kids = n.kids
    if near:
        e = [slab(k, o, inv, 0.001, st["t"])[1] for k in kids]
        if e[1] < e[0]: kids = (kids[1], kids[0])
    for k in kids: walk(k, o, d, inv, st, near)
tot = {False: [0, 0], True: [0, 0]}; hits = 0; nr = 400
for _ in range(nr):
    # rays from outside aimed into the cluster
    o = rng.normal(size=3); o = 82.5 + 400 * o / np.linalg.norm(o)
    tgt = rng.uniform(20, 145, size=3); d = tgt - o
    inv = 1.0 / d
    res = {}
    for near in (False, True):
        st = {"box": 0, "sph": 0, "t": np.inf}
        walk(root, o, d, inv, st, near)
        tot[near][0] += st["box"]; tot[near][1] += st["sph"]; res[near] = st["t"]
    assert res[False] == res[True]
    hits += res[False] < np.inf
print("rays", nr, "hit", hits)
for near in (False, True):
    print("near-first" if near else "fixed order", "boxes/ray", tot[near][0]/nr, "sphere tests/ray", tot[near][1]/nr)

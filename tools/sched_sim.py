#!/usr/bin/env python3
"""Offline study of the traversal kernel's wave scheduling.

Records, with the host build of the device core, the sequence of program units each ray of the final scene steps
through, groups rays into waves of 64 and replays scheduling policies under a simple instruction-cost model.
Prints VALU-instruction estimates per wave-segment for each policy.  Test/tuning tool, not part of the product.
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import raytracing_rust_amd as rt  # noqa: E402
from raytracing_rust_amd import _capi, scenes  # noqa: E402
import helpers as H  # noqa: E402

BOX, ENTER, EXIT, MEDIUM, NOP, END, SYNC = 0, 3, 4, 5, 6, 7, 10
CL = {33: "SPHERE_S", 34: "QUAD_AA", 35: "SPHERE", 36: "QUAD", 37: "MEDEVAL", 38: "SYNC", 39: "BOX6"}
COST = {BOX: 45, ENTER: 250, EXIT: 250, MEDIUM: 25, NOP: 10, END: 25, 33: 110, 34: 400, 35: 130, 36: 150, 37: 350, 38: 10, 39: 330}
TRIP_OVERHEAD = 55


def trace(desc, rays, cap=512):
    lib = H.hostsim_lib()
    lib.sim_trace_steps.argtypes = [C.POINTER(_capi.SceneDesc), H.PD, C.c_size_t, C.c_uint32, C.POINTER(C.c_uint8), C.POINTER(C.c_uint32)]
    n = len(rays)
    seq = np.zeros((n, cap), dtype=np.uint8)
    ln = np.zeros(n, dtype=np.uint32)
    r = np.ascontiguousarray(rays)
    lib.sim_trace_steps(C.byref(desc), H.dptr(r), n, cap, seq.ctypes.data_as(C.POINTER(C.c_uint8)), ln.ctypes.data_as(C.POINTER(C.c_uint32)))
    assert ln.max() < cap
    return [seq[i, :ln[i]] for i in range(n)]


def simulate(wave, policy):
    """wave: list of 64 sequences.  Returns (cost, trips per class)."""
    ptr = [0] * len(wave)
    cost = 0
    trips = {}
    while True:
        cur = [w[p] if p < len(w) else None for w, p in zip(wave, ptr)]
        live = [c for c in cur if c is not None]
        if not live:
            break
        counts = {}
        for c in live:
            counts[c] = counts.get(c, 0) + 1
        pick = policy(counts, len(live))
        for k, c in enumerate(cur):
            if c == pick:
                ptr[k] += 1
        cost += COST[pick] + TRIP_OVERHEAD
        trips[pick] = trips.get(pick, 0) + 1
    return cost, trips


CHEAP = (BOX, ENTER, EXIT, MEDIUM, NOP, END)


def pol_current(counts, nlive):
    if BOX in counts:
        return BOX
    for c in (END, ENTER, EXIT, MEDIUM, NOP):
        if c in counts:
            return c
    leaf = {c: n for c, n in counts.items() if c != 38}
    if not leaf:
        return 38
    return max(leaf, key=lambda c: leaf[c])


def pol_majority_weighted(counts, nlive):
    """Everything competes; score = lanes / cost; SYNC only when alone."""
    cand = {c: n for c, n in counts.items() if c != 38}
    if not cand:
        return 38
    return max(cand, key=lambda c: cand[c] / (COST[c] + TRIP_OVERHEAD))


def pol_majority_plain(counts, nlive):
    cand = {c: n for c, n in counts.items() if c != 38}
    if not cand:
        return 38
    return max(cand, key=lambda c: cand[c])


def pol_box_threshold(th):
    def pol(counts, nlive):
        cand = {c: n for c, n in counts.items() if c != 38}
        if not cand:
            return 38
        if counts.get(BOX, 0) >= th * nlive or set(cand) == {BOX}:
            return BOX
        nb = {c: n for c, n in cand.items() if c != BOX}
        for c in (END, ENTER, EXIT, MEDIUM, NOP):
            if c in nb:
                return c
        return max(nb, key=lambda c: nb[c]) if nb else BOX
    return pol


def report(name, seqs, group):
    waves = [seqs[i:i + 64] for i in range(0, len(seqs) - 63, 64)] if group == "order" else None
    if group == "random":
        idx = np.random.default_rng(1).permutation(len(seqs))
        waves = [[seqs[j] for j in idx[i:i + 64]] for i in range(0, len(seqs) - 63, 64)]
    per_lane = np.mean([sum(COST[c] for c in s) for s in seqs])
    mx = np.mean([max(sum(COST[c] for c in s) for s in w) for w in waves])
    print(f"== {name} / {group}: {len(waves)} waves; converged ideal {per_lane:.0f}, mean of per-wave max lane {mx:.0f}, mean steps/lane {np.mean([len(s) for s in seqs]):.1f}")
    pols = {"current": pol_current, "majority/cost": pol_majority_weighted, "majority": pol_majority_plain,
            "box>=25%": pol_box_threshold(0.25), "box>=50%": pol_box_threshold(0.5)}
    for pn, pol in pols.items():
        res = [simulate(w, pol) for w in waves[:60]]
        cost = np.mean([r[0] for r in res])
        trips = {}
        for r in res:
            for c, n in r[1].items():
                trips[c] = trips.get(c, 0) + n / len(res)
        tr = " ".join(f"{CL.get(c, {0:'BOX',3:'ENT',4:'EXT',5:'MED',6:'NOP',7:'END'}.get(c, c))}:{n:.0f}" for c, n in sorted(trips.items()))
        print(f"   {pn:14s} cost {cost:8.0f}   trips {tr}")


def main():
    g = rt.Graph()
    w, l, cam = scenes.build("final", g, width=128, spp=100, depth=75)
    desc = rt.flatten(g, w, l)
    # primary rays in 8x8 tiles
    rays = rt.pixel_center_rays(cam, time=0.3).reshape(cam.image_height, cam.image_width, 7)
    tiles = rays.reshape(cam.image_height // 8, 8, cam.image_width // 8, 8, 7).transpose(0, 2, 1, 3, 4).reshape(-1, 7)
    prim = trace(desc, tiles)
    report("primary", prim, "order")
    # secondary rays: cosine-ish directions about the first-hit normal
    hit = H.hostsim_first_hit(desc, tiles)
    ok = np.isfinite(hit["t"])
    rng = np.random.default_rng(7)
    p = tiles[:, 0:3] + tiles[:, 3:6] * np.where(ok, hit["t"], 1.0)[:, None]
    d = hit["normal"] + rng.normal(size=(len(tiles), 3)) * 0.8
    d[~ok] = rng.normal(size=(int((~ok).sum()), 3))
    sec = np.concatenate([p, d, tiles[:, 6:7]], axis=1)[ok]
    secs = trace(desc, sec)
    report("secondary", secs, "order")
    report("secondary", secs, "random")


if __name__ == "__main__" and not any(a in sys.argv for a in ("--refill", "--sort", "--phased", "--threshold", "--exits", "--ideal")):
    main()


def simulate_refill(seqs, policy, lanes=64, sync_as_nop=True):
    """Persistent wave: a lane that finishes its ray takes the next one from the pool.  Returns cost per ray."""
    pool = list(seqs)
    nxt = 0
    cur_seq = [None] * lanes
    ptr = [0] * lanes
    for k in range(lanes):
        if nxt < len(pool):
            cur_seq[k] = pool[nxt]; nxt += 1
    cost = 0
    trips = {}
    while True:
        cur = []
        for k in range(lanes):
            while cur_seq[k] is not None and ptr[k] >= len(cur_seq[k]):
                if nxt < len(pool):
                    cur_seq[k] = pool[nxt]; nxt += 1; ptr[k] = 0
                else:
                    cur_seq[k] = None
            c = cur_seq[k][ptr[k]] if cur_seq[k] is not None else None
            if sync_as_nop and c == 38:
                c = NOP
            cur.append(c)
        live = [c for c in cur if c is not None]
        if not live:
            break
        counts = {}
        for c in live:
            counts[c] = counts.get(c, 0) + 1
        pick = policy(counts, len(live))
        for k, c in enumerate(cur):
            if c == pick:
                ptr[k] += 1
        cost += COST[pick] + TRIP_OVERHEAD
        trips[pick] = trips.get(pick, 0) + 1
    return cost / len(pool) * 64, {c: n / len(pool) * 64 for c, n in trips.items()}


def report_refill(name, seqs):
    print(f"== {name} / persistent wave with refill (cost per 64 rays)")
    for pn, pol in {"current": pol_current, "majority/cost": pol_majority_weighted, "majority": pol_majority_plain}.items():
        cost, trips = simulate_refill(seqs[:64 * 40], pol)
        tr = " ".join(f"{CL.get(c, {0:'BOX',3:'ENT',4:'EXT',5:'MED',6:'NOP',7:'END'}.get(c, c))}:{n:.0f}" for c, n in sorted(trips.items()))
        print(f"   {pn:14s} cost {cost:8.0f}   trips {tr}")


if __name__ == "__main__" and "--refill" in sys.argv and "--sort" not in sys.argv:
    g = rt.Graph()
    w, l, cam = scenes.build("final", g, width=128, spp=100, depth=75)
    desc = rt.flatten(g, w, l)
    rays = rt.pixel_center_rays(cam, time=0.3).reshape(cam.image_height, cam.image_width, 7)
    tiles = rays.reshape(cam.image_height // 8, 8, cam.image_width // 8, 8, 7).transpose(0, 2, 1, 3, 4).reshape(-1, 7)
    hit = H.hostsim_first_hit(desc, tiles)
    ok = np.isfinite(hit["t"])
    rng = np.random.default_rng(7)
    p = tiles[:, 0:3] + tiles[:, 3:6] * np.where(ok, hit["t"], 1.0)[:, None]
    d = hit["normal"] + rng.normal(size=(len(tiles), 3)) * 0.8
    sec = np.concatenate([p, d, tiles[:, 6:7]], axis=1)[ok]
    secs = trace(desc, sec)
    report_refill("secondary", secs)


def morton3(ix, iy, iz, bits=5):
    code = np.zeros_like(ix)
    for b in range(bits):
        code |= ((ix >> b) & 1) << (3 * b) | ((iy >> b) & 1) << (3 * b + 1) | ((iz >> b) & 1) << (3 * b + 2)
    return code


if __name__ == "__main__" and "--sort" in sys.argv:
    g = rt.Graph()
    w, l, cam = scenes.build("final", g, width=192, spp=100, depth=75)
    desc = rt.flatten(g, w, l)
    rays = rt.pixel_center_rays(cam, time=0.3)
    hit = H.hostsim_first_hit(desc, rays)
    ok = np.isfinite(hit["t"])
    rng = np.random.default_rng(7)
    p = rays[:, 0:3] + rays[:, 3:6] * np.where(ok, hit["t"], 1.0)[:, None]
    d = hit["normal"] + rng.normal(size=(len(rays), 3)) * 0.8
    sec = np.concatenate([p, d, rays[:, 6:7]], axis=1)[ok]
    sec = sec[rng.permutation(len(sec))][:64 * 120]
    secs = trace(desc, sec)
    def run(order, name):
        s2 = [secs[i] for i in order]
        waves = [s2[i:i + 64] for i in range(0, len(s2) - 63, 64)]
        cost = np.mean([simulate(wv, pol_majority_plain)[0] for wv in waves[:80]])
        cost_r, _ = simulate_refill(s2[:64 * 60], pol_majority_plain)
        print(f"   {name:34s} majority {cost:8.0f}   majority+refill {cost_r:8.0f}")
    n = len(secs)
    print("== secondary rays, wave grouping by sort key")
    run(np.arange(n), "random")
    o = sec[:, 0:3]; dd = sec[:, 3:6]
    octant = (dd[:, 0] > 0).astype(np.int64) | ((dd[:, 1] > 0).astype(np.int64) << 1) | ((dd[:, 2] > 0).astype(np.int64) << 2)
    cell = np.clip(((o + 1000.0) / 2000.0 * 32).astype(np.int64), 0, 31)
    mort = morton3(cell[:, 0], cell[:, 1], cell[:, 2])
    run(np.argsort(octant, kind="stable"), "direction octant")
    run(np.argsort(mort, kind="stable"), "origin morton (32^3)")
    run(np.argsort(octant * (1 << 15) + mort, kind="stable"), "octant, then origin morton")
    run(np.argsort(mort * 8 + octant, kind="stable"), "origin morton, then octant")
    ln = np.array([len(s) for s in secs])
    run(np.argsort(ln, kind="stable"), "oracle: by sequence length")


class PolPhased:
    """Two alternating phases: walk boxes until no lane stands on one, then drain every non-box unit until no lane
    stands on one (largest class first), and so on.  SYNC is passed only when nothing else is left."""
    def __init__(self):
        self.phase = "box"

    def __call__(self, counts, nlive):
        cand = {c: n for c, n in counts.items() if c != 38}
        if not cand:
            return 38
        if self.phase == "box":
            if BOX in cand:
                return BOX
            self.phase = "leaf"
        nb = {c: n for c, n in cand.items() if c != BOX}
        if not nb:
            self.phase = "box"
            return BOX
        for c in (END, ENTER, EXIT, MEDIUM, NOP):
            if c in nb:
                return c
        return max(nb, key=lambda c: nb[c])


if __name__ == "__main__" and "--phased" in sys.argv:
    g = rt.Graph()
    w, l, cam = scenes.build("final", g, width=128, spp=100, depth=75)
    desc = rt.flatten(g, w, l)
    rays = rt.pixel_center_rays(cam, time=0.3).reshape(cam.image_height, cam.image_width, 7)
    tiles = rays.reshape(cam.image_height // 8, 8, cam.image_width // 8, 8, 7).transpose(0, 2, 1, 3, 4).reshape(-1, 7)
    hit = H.hostsim_first_hit(desc, tiles)
    ok = np.isfinite(hit["t"])
    rng = np.random.default_rng(7)
    p = tiles[:, 0:3] + tiles[:, 3:6] * np.where(ok, hit["t"], 1.0)[:, None]
    d = hit["normal"] + rng.normal(size=(len(tiles), 3)) * 0.8
    sec = np.concatenate([p, d, tiles[:, 6:7]], axis=1)[ok]
    secs = trace(desc, sec)
    waves = [secs[i:i + 64] for i in range(0, len(secs) - 63, 64)]
    for pn, mk in {"current": lambda: pol_current, "phased": PolPhased}.items():
        res = [simulate(wv, mk()) for wv in waves[:60]]
        trips = {}
        for r in res:
            for c, n in r[1].items():
                trips[c] = trips.get(c, 0) + n / len(res)
        tr = " ".join(f"{CL.get(c, {0:'BOX',3:'ENT',4:'EXT',5:'MED',6:'NOP',7:'END'}.get(c, c))}:{n:.0f}" for c, n in sorted(trips.items()))
        print(f"   {pn:14s} cost {np.mean([r[0] for r in res]):8.0f}   trips {tr}")
    nb = np.array([int((s == BOX).sum()) for s in secs])
    mx = np.array([max(int((s == BOX).sum()) for s in wv) for wv in waves])
    print("box steps per lane: mean %.1f median %.0f p90 %.0f max %d; mean of per-wave max %.1f" % (nb.mean(), np.median(nb), np.percentile(nb, 90), nb.max(), mx.mean()))
    ln = np.array([len(s) for s in secs])
    print("all steps per lane: mean %.1f p90 %.0f max %d" % (ln.mean(), np.percentile(ln, 90), ln.max()))


# Costs measured on the GPU (tools/prof_trips.py, wall cycles per trip with 4 waves per SIMD)
MEASURED = {BOX: 772, ENTER: 2185, EXIT: 2185, MEDIUM: 2185, NOP: 2185, END: 2185, 33: 3191, 34: 3853, 35: 3083, 36: 3853, 37: 7296, 38: 1796, 39: 4801}


def simulate_threshold_refill(seqs, lanes=64, threshold=32, sync_as_nop=True, refill_cost=3000):
    """Persistent wave, phased policy; finished lanes wait until `threshold` lanes are idle (or nothing runs), then
    all idle lanes take new rays in one refill event."""
    pool = list(seqs)
    nxt = 0
    cur_seq = [None] * lanes
    ptr = [0] * lanes
    cost = 0
    trips = {}
    pol = PolPhased()
    while True:
        idle = [k for k in range(lanes) if cur_seq[k] is None or ptr[k] >= len(cur_seq[k])]
        running = lanes - len(idle)
        if nxt < len(pool) and (len(idle) >= threshold or running == 0):
            for k in idle:
                if nxt < len(pool):
                    cur_seq[k] = pool[nxt]; nxt += 1; ptr[k] = 0
            cost += refill_cost
            trips["refill"] = trips.get("refill", 0) + 1
            continue
        cur = []
        for k in range(lanes):
            c = cur_seq[k][ptr[k]] if cur_seq[k] is not None and ptr[k] < len(cur_seq[k]) else None
            if sync_as_nop and c == 38:
                c = NOP
            cur.append(c)
        live = [c for c in cur if c is not None]
        if not live:
            break
        counts = {}
        for c in live:
            counts[c] = counts.get(c, 0) + 1
        pick = pol(counts, len(live))
        for k, c in enumerate(cur):
            if c == pick:
                ptr[k] += 1
        cost += MEASURED[pick]
        trips[pick] = trips.get(pick, 0) + 1
    return cost / len(pool) * 64, {c: n / len(pool) * 64 for c, n in trips.items()}


if __name__ == "__main__" and "--threshold" in sys.argv:
    g = rt.Graph()
    w, l, cam = scenes.build("final", g, width=128, spp=100, depth=75)
    desc = rt.flatten(g, w, l)
    rays = rt.pixel_center_rays(cam, time=0.3).reshape(cam.image_height, cam.image_width, 7)
    tiles = rays.reshape(cam.image_height // 8, 8, cam.image_width // 8, 8, 7).transpose(0, 2, 1, 3, 4).reshape(-1, 7)
    hit = H.hostsim_first_hit(desc, tiles)
    ok = np.isfinite(hit["t"])
    rng = np.random.default_rng(7)
    p = tiles[:, 0:3] + tiles[:, 3:6] * np.where(ok, hit["t"], 1.0)[:, None]
    d = hit["normal"] + rng.normal(size=(len(tiles), 3)) * 0.8
    sec = np.concatenate([p, d, tiles[:, 6:7]], axis=1)[ok]
    for label, seqs in (("primary", trace(desc, tiles)), ("secondary", trace(desc, sec))):
        seqs = seqs[:64 * 100]
        waves = [seqs[i:i + 64] for i in range(0, len(seqs) - 63, 64)]
        COST.update(MEASURED); globals()['TRIP_OVERHEAD'] = 0
        base = np.mean([simulate(wv, PolPhased())[0] for wv in waves])
        print(f"{label}: phased, no refill: {base:.0f} cycles per 64 rays")
        for th in (64, 48, 32, 16, 8, 1):
            c, tr = simulate_threshold_refill(seqs, threshold=th)
            print(f"   refill at >= {th:2d} idle lanes: {c:8.0f}   refills {tr.get('refill', 0):.1f} box trips {tr.get(BOX, 0):.0f}")


# ---- phased policy with early exits (costs: tools/prof_trips.py of the current kernels)
MEASURED2 = {BOX: 796, ENTER: 2008, EXIT: 2008, MEDIUM: 2008, NOP: 2008, END: 2008, 13: 2008, 33: 3256, 34: 3113, 35: 3750, 36: 3853,
             37: 6882, 38: 1904, 39: 4885, 41: 6882}


class PolPhasedExit:
    """Phased policy; the box phase ends early once at most `t_box` lanes stand on a box while at least `u_leaf` lanes
    wait on one leaf class; the leaf phase runs classes largest first and, while boxes are pending, skips classes with
    fewer than `u_min` lanes."""
    def __init__(self, t_box=0, u_leaf=65, u_min=0):
        self.phase = "box"; self.t_box = t_box; self.u_leaf = u_leaf; self.u_min = u_min

    def __call__(self, counts, nlive):
        cand = {c: n for c, n in counts.items() if c != 38}
        if not cand:
            return 38
        nb = {c: n for c, n in cand.items() if c != BOX}
        if self.phase == "box":
            n_box = cand.get(BOX, 0)
            big = max(nb.values()) if nb else 0
            if n_box > 0 and not (n_box <= self.t_box and big >= self.u_leaf):
                return BOX
            self.phase = "leaf"
        if not nb:
            self.phase = "box"
            return BOX
        for c in (END, ENTER, EXIT, MEDIUM, NOP, 13):
            if c in nb:
                return c
        c = max(nb, key=lambda c: nb[c])
        if BOX in cand and nb[c] < self.u_min:
            self.phase = "box"
            return BOX
        return c


if __name__ == "__main__" and "--exits" in sys.argv:
    g = rt.Graph()
    w, l, cam = scenes.build("final", g, width=128, spp=100, depth=75)
    desc = rt.flatten(g, w, l)
    rays = rt.pixel_center_rays(cam, time=0.3).reshape(cam.image_height, cam.image_width, 7)
    tiles = rays.reshape(cam.image_height // 8, 8, cam.image_width // 8, 8, 7).transpose(0, 2, 1, 3, 4).reshape(-1, 7)
    hit = H.hostsim_first_hit(desc, tiles)
    ok = np.isfinite(hit["t"])
    rng = np.random.default_rng(7)
    p = tiles[:, 0:3] + tiles[:, 3:6] * np.where(ok, hit["t"], 1.0)[:, None]
    d = hit["normal"] + rng.normal(size=(len(tiles), 3)) * 0.8
    sec = np.concatenate([p, d, tiles[:, 6:7]], axis=1)[ok]
    COST.update(MEASURED2); globals()['TRIP_OVERHEAD'] = 0
    for label, seqs in (("primary", trace(desc, tiles)), ("secondary", trace(desc, sec))):
        # the world-program kernel sees the sequence up to the DEFER unit and what follows the section's END
        main_part, tail_part = [], []
        for s in seqs[:64 * 150]:
            at = np.nonzero(s == 13)[0]
            if len(at):
                e = at[0] + 1 + int(np.nonzero(s[at[0] + 1:] == END)[0][0])
                main_part.append(np.concatenate([s[:at[0] + 1], s[e + 1:]])); tail_part.append(s[at[0] + 1:e + 1])
            else:
                main_part.append(s)
        for part, ss in (("world program", main_part), ("deferred section", tail_part)):
            waves = [ss[i:i + 64] for i in range(0, len(ss) - 63, 64)]
            if not waves:
                continue
            print(f"{label} / {part}: {len(waves)} waves")
            for name, mk in [("phased (current)", lambda: PolPhasedExit())] + \
                    [(f"box exit t<={t} u>={u}", (lambda t=t, u=u: PolPhasedExit(t, u))) for t in (4, 8, 16) for u in (16, 32)] + \
                    [(f"leaf min {m}", (lambda m=m: PolPhasedExit(0, 65, m))) for m in (4, 8, 16)] + \
                    [(f"exit t<=8 u>=32 + leaf min 8", lambda: PolPhasedExit(8, 32, 8))]:
                res = [simulate(wv, mk()) for wv in waves]
                trips = {}
                for r in res:
                    for c, n in r[1].items():
                        trips[c] = trips.get(c, 0) + n / len(res)
                print(f"   {name:30s} {np.mean([r[0] for r in res]):8.0f} cycles/64 rays   box trips {trips.get(BOX, 0):5.1f} BOX6 {trips.get(39, 0):4.1f} SPHERE_S {trips.get(33, 0):4.1f}")


if __name__ == "__main__" and "--ideal" in sys.argv:
    g = rt.Graph()
    w, l, cam = scenes.build("final", g, width=128, spp=100, depth=75)
    desc = rt.flatten(g, w, l)
    rays = rt.pixel_center_rays(cam, time=0.3).reshape(cam.image_height, cam.image_width, 7)
    tiles = rays.reshape(cam.image_height // 8, 8, cam.image_width // 8, 8, 7).transpose(0, 2, 1, 3, 4).reshape(-1, 7)
    hit = H.hostsim_first_hit(desc, tiles)
    ok = np.isfinite(hit["t"])
    rng = np.random.default_rng(7)
    p = tiles[:, 0:3] + tiles[:, 3:6] * np.where(ok, hit["t"], 1.0)[:, None]
    d = hit["normal"] + rng.normal(size=(len(tiles), 3)) * 0.8
    sec = np.concatenate([p, d, tiles[:, 6:7]], axis=1)[ok]
    seqs = trace(desc, sec)[:64 * 150]
    tot = {}
    for s in seqs:
        for c in s:
            tot[int(c)] = tot.get(int(c), 0) + 1
    n = len(seqs)
    print("steps per ray by kind:", {k: round(v / n, 2) for k, v in sorted(tot.items())})
    print("perfectly packed waves: %.0f cycles per 64 rays" % (sum(MEASURED2[k] * v for k, v in tot.items()) / n))

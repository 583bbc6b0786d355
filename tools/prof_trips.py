#!/usr/bin/env python3
"""Cycle accounting of the traversal loop on the GPU (tuning build: make -C raytracing_rust_amd/csrc prof).

usage (on the GPU box): RTX_AMD_LIB=tools/_build/librtx_amd_prof.so python3 tools/prof_trips.py [scene]
Prints, per trip class of closest_hit: share of wave cycles, trips per wave-segment, mean cycles per trip and
mean lanes served per trip.  Test/tuning tool, not part of the product.
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import raytracing_rust_amd as rt  # noqa: E402
from raytracing_rust_amd import _capi  # noqa: E402

NAMES = {0: "cheap", 1: "SPHERE_S", 2: "QUAD_AA", 3: "SPHERE", 4: "QUAD", 5: "MEDIUM", 6: "SYNC / sweep: rest", 7: "BOX6",
         8: "box loop exit", 9: "box step"}
# In a kernel that sweeps the flat stretch of the program (walk_flat), classes 1 to 5 and 7 are units of the sweep
# (one trip = one unit for the whole wave), 6 its boxes, transforms and the rest; 0, 8, 9 (and 7 inside the head
# section) belong to the per-ray walk of the new camera rays through the head section.


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "final"
    lib = _capi.lib
    lib.rtx_debug_prof_read.argtypes = [C.POINTER(C.c_uint64), C.c_int]
    ctx = rt.Context(0)
    g = rt.Graph()
    world, lights, cam = rt.scenes.build(name, g, width=800, spp=5000, depth=75)
    scene = rt.Scene.from_graph(ctx, g, world, lights)
    accum = torch.zeros(cam.image_width * cam.image_height * 3, dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    buf = (C.c_uint64 * 64)()
    lib.rtx_debug_prof_read(buf, 1)
    ctx.reset_stats()
    opts = rt.render_opts(cam, sample_begin=0, sample_end=16, seed=rt.scenes.PATH_SEED)
    scene.render_device(cam, opts, accum, stream)
    torch.cuda.synchronize()
    lib.rtx_debug_prof_read(buf, 0)
    st = ctx.stats()
    wsegs = st["segments"] / 64.0
    total = sum(buf[k] for k in range(16))
    print(f"{name}: {st['segments']} segments, {total / wsegs:.0f} accounted cycles per wave-segment")
    print(f"{'class':14s} {'share':>6s} {'trips/wseg':>10s} {'cyc/trip':>9s} {'lanes/trip':>10s}")
    for k in range(16):
        if buf[16 + k] == 0:
            continue
        print(f"{NAMES.get(k, k):14s} {buf[k] / total:6.1%} {buf[16 + k] / wsegs:10.1f} {buf[k] / buf[16 + k]:9.0f} {buf[32 + k] / buf[16 + k]:10.1f}")


if __name__ == "__main__":
    main()

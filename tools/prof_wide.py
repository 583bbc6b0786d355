#!/usr/bin/env python3
"""Trip accounting of the folded-table walk (walk_stream_wide) on the stress scene (tuning build: make -C
raytracing_rust_amd/csrc prof).
usage (GPU box): RTX_AMD_LIB=tools/_build/librtx_amd_prof.so python3 tools/prof_wide.py [--prims N] [--samples S]"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import raytracing_rust_amd as rt  # noqa: E402
from raytracing_rust_amd import _capi  # noqa: E402


def main():
    import hiprt
    ap = argparse.ArgumentParser()
    ap.add_argument("--prims", type=int, default=1_000_000)
    ap.add_argument("--samples", type=int, default=4)
    args = ap.parse_args()
    lib = _capi.lib
    lib.rtx_debug_wprof_read.argtypes = [C.POINTER(C.c_uint64), C.c_int]
    ctx = rt.Context(0)
    g = rt.Graph()
    g.use_context(ctx)
    world, lights, cam = rt.scenes.build("stress", g, n_prims=args.prims)
    scene = rt.Scene.from_graph(ctx, g, world, lights)
    acc = hiprt.DeviceArray(cam.image_width * cam.image_height * 3)
    buf = (C.c_uint64 * 32)()
    opts = rt.render_opts(cam, sample_begin=0, sample_end=args.samples, seed=rt.scenes.PATH_SEED)
    scene.render_device(cam, opts, acc.ptr)
    hiprt.synchronize()
    lib.rtx_debug_wprof_read(buf, 1)
    ctx.reset_stats()
    scene.render_device(cam, opts, acc.ptr)
    hiprt.synchronize()
    lib.rtx_debug_wprof_read(buf, 0)
    st = ctx.stats()
    b = [int(x) for x in buf]
    trips = max(1, b[0])
    print(f"stress {args.prims} primitives, {st['segments']} segments; trips: {trips}")
    print(f"lanes with a ray per trip {b[1] / trips:.1f}; waiting for a record step {b[2] / trips:.1f}, a leaf step {b[3] / trips:.1f}, a quad step {b[4] / trips:.1f}")
    for name, t, n in (("record", 5, 6), ("leaf", 7, 8), ("quad", 9, 10)):
        print(f"{name:7s} step ran in {b[t] / trips:6.1%} of the trips with {b[n] / max(1, b[t]):5.1f} lanes")
    print(f"trips with records from memory: {b[11] / trips:.2%}")
    tot = max(1, b[17] + b[18])
    for name, k in (("pops, hand-in, refill", 18), ("issue of the reads", 12), ("record step", 13), ("leaf step", 14), ("quad step", 15), ("records from memory", 16)):
        print(f"{name:20s} {b[k] / tot:6.1%} of the trip cycles, {b[k] / trips:7.0f} cycles per trip")
    print(f"cycles per trip: {tot / trips:.0f}; lane steps per trip {(b[6] + b[8] + b[10]) / trips:.1f}")
    scene.close()
    ctx.close()


if __name__ == "__main__":
    main()

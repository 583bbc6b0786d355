#!/usr/bin/env python3
"""Cycle accounting of the shading kernel's sections on the GPU (tuning build: make -C raytracing_rust_amd/csrc prof).
usage (GPU box): RTX_AMD_LIB=tools/_build/librtx_amd_prof.so python3 tools/prof_shade.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import raytracing_rust_amd as rt  # noqa: E402
from raytracing_rust_amd import _capi  # noqa: E402

NAMES = {1: "miss", 2: "material record", 3: "resolve_hit", 4: "emission", 5: "metal", 6: "dielectric",
         7: "texture_value", 8: "direction sampling", 9: "pdf (lights)", 10: "throughput + roulette", 15: "(wait for other branches)"}
PHASES = {12: "sort phase (per wave and trip)", 13: "shading pass (per wave and pass)",
          14: "traversal: pass start to walk start", 11: "traversal: pass start to hit stored"}


def main():
    lib = _capi.lib
    lib.rtx_debug_sprof_read.argtypes = [C.POINTER(C.c_uint64), C.c_int]
    ctx = rt.Context(0)
    g = rt.Graph()
    world, lights, cam = rt.scenes.build("final", g, width=800, spp=5000, depth=75)
    scene = rt.Scene.from_graph(ctx, g, world, lights)
    accum = torch.zeros(cam.image_width * cam.image_height * 3, dtype=torch.float64, device="cuda")
    buf = (C.c_uint64 * 32)()
    lib.rtx_debug_sprof_read(buf, 1)
    ctx.reset_stats()
    scene.render_device(cam, rt.render_opts(cam, sample_begin=0, sample_end=16, seed=rt.scenes.PATH_SEED), accum,
                        torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    lib.rtx_debug_sprof_read(buf, 0)
    st = ctx.stats()
    wsegs = st["segments"] / 64.0 / 8.0          # one block in eight is measured
    total = sum(buf[k] for k in range(12))
    print(f"{st['segments']} segments; {total / wsegs:.0f} accounted cycles per wave (without the wait bucket)")
    print(f"{'section':28s} {'share':>6s} {'waves that ran it':>18s} {'cycles each':>12s}")
    for k in PHASES:
        if buf[16 + k]:
            print(f"{PHASES[k]:34s} {buf[16 + k]} times, {buf[k] / buf[16 + k]:10.0f} cycles each, {buf[k] / wsegs:10.0f} cycles per 64 segments")
    for k in list(range(12)) + [15]:
        if buf[16 + k]:
            print(f"{str(NAMES.get(k, k)):28s} {buf[k] / total:6.1%} {buf[16 + k] / wsegs:18.2f} {buf[k] / buf[16 + k]:12.0f}")


if __name__ == "__main__":
    main()

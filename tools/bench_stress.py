#!/usr/bin/env python3
"""Throughput on the synthetic stress configuration: 1 M random spheres and quads in one BVH, 3840x2160, depth 50.

    python tools/bench_stress.py [--prims N] [--width W] [--samples S] [--steps K]

A step renders S of the 4096 stratified sample indices of the whole frame.  The hit program of this scene (about
70 MB) is walked from global memory; the numbers say how the path behaves once the scene no longer fits in LDS.
Prints one JSON line.  Not the headline benchmark (that is bench.py at the repository root).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prims", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--samples", type=int, default=4)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--depth", type=int, default=50)
    ap.add_argument("--pipeline", type=int, default=0)
    ap.add_argument("--gpu-build", action="store_true", help="sort the BVH on the GPU (rtx_bvh_leaf_order)")
    args = ap.parse_args()
    import torch
    import raytracing_rust_amd as rt
    if not torch.cuda.is_available():
        raise SystemExit("needs a GPU: the backend has no CPU path")
    ctx = rt.Context(0)
    g = rt.Graph()
    if args.gpu_build:
        g.use_context(ctx)
    t0 = time.perf_counter()
    world, lights, cam = rt.scenes.build("stress", g, width=args.width, depth=args.depth, n_prims=args.prims)
    t1 = time.perf_counter()
    scene = rt.Scene.from_graph(ctx, g, world, lights)
    t2 = time.perf_counter()
    npix = cam.image_width * cam.image_height
    accum = torch.zeros(npix * 3, dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def step(k):
        opts = rt.render_opts(cam, sample_begin=k * args.samples, sample_end=(k + 1) * args.samples,
                              seed=rt.scenes.PATH_SEED, pipeline=args.pipeline)
        scene.render_device(cam, opts, accum, stream)

    step(0)
    torch.cuda.synchronize()
    ctx.reset_stats()
    t3 = time.perf_counter()
    for k in range(args.steps):
        step(1 + k)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t3
    st = ctx.stats()
    assert bool(torch.isfinite(accum).all())
    print(json.dumps({
        "workload": f"stress scene, {args.prims} primitives, {cam.image_width}x{cam.image_height}, depth {args.depth}, "
                    f"{args.samples} sample indices per step",
        "Msamples_per_s": st["paths"] / dt / 1e6, "Msegments_per_s": st["segments"] / dt / 1e6,
        "segments_per_path": st["segments"] / max(1, st["paths"]), "ms_per_step": dt / args.steps * 1e3,
        "scene_build_s": t1 - t0, "bvh_built_on": "gpu" if args.gpu_build else "host", "flatten_compile_upload_s": t2 - t1,
        "traverse_ms_per_launch": st["traverse_ms"] / max(1, st["traverse_launches"]),
        "traverse_share_of_render_time": st["traverse_ms"] / max(1e-9, st["render_ms"]),
        "mean_radiance": float(accum.mean()) / ((args.steps + 1) * args.samples)}), flush=True)
    scene.close()
    ctx.close()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Throughput of every configuration the path is quoted on, GPU next to the oracle on the host cores.

    python tools/bench_configs.py [--no-cpu] [--out gpurun_out/configs.json]

quick   final scene 400x400, 100 samples, depth 20        (src/main.rs:31-39)   whole job
cornell Cornell box + glass sphere 600x600, 961, depth 50   (src/main.rs:13-20)   whole job
smoke   Cornell box with two smoke boxes 600x600, 961, 50   (authored)            whole job
final   final scene 800x800, 4900 samples, depth 75         (src/main.rs:22-29)   whole job
stress  1 M spheres/quads in one BVH, 3840x2160, depth 50   (synthetic)           8 of the 4096 sample indices

The GPU renders the whole job once untimed (warm-up: workspace allocation) and once timed, through
rtx_render_device with the accumulator resident in HBM.  The oracle (test infrastructure: a CPU restatement of the
reference) is timed on a bounded number of sample indices of the same frame; its cost is linear in samples
(camera.rs:259-267).  Prints one JSON object per configuration.  Not the headline benchmark (bench.py is).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

# name -> (GPU sample indices or None = all, oracle sample indices, stress primitives)
CONFIGS = {
    "quick": (None, 100, None),
    "cornell": (None, 64, None),
    "smoke": (None, 64, None),
    "final": (None, 64, None),
    "stress": (8, 1, 1_000_000),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--only", default="")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import torch
    import raytracing_rust_amd as rt
    if not torch.cuda.is_available():
        raise SystemExit("needs a GPU: the backend has no CPU path")
    ctx = rt.Context(0)
    threads = min(16, os.cpu_count() or 1)
    results = []
    for name, (gpu_samples, cpu_samples, prims) in CONFIGS.items():
        if args.only and name not in args.only.split(","):
            continue
        g = rt.Graph()
        kw = {"n_prims": prims} if prims else {}
        world, lights, cam = rt.scenes.build(name, g, **kw)
        scene = rt.Scene.from_graph(ctx, g, world, lights)
        total = cam.sqrt_spp ** 2
        n = total if gpu_samples is None else gpu_samples
        npix = cam.image_width * cam.image_height
        accum = torch.zeros(npix * 3, dtype=torch.float64, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        for timed in (False, True):
            accum.zero_()
            torch.cuda.synchronize()
            ctx.reset_stats()
            t0 = time.perf_counter()
            scene.render_device(cam, rt.render_opts(cam, sample_begin=0, sample_end=n, seed=rt.scenes.PATH_SEED + (1 if timed else 0)), accum, stream)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        st = ctx.stats()
        assert bool(torch.isfinite(accum).all()) and st["paths"] == npix * n
        r = {"config": name, "frame": f"{cam.image_width}x{cam.image_height}", "samples_per_pixel_rendered": n,
             "samples_per_pixel_of_the_job": total, "max_depth": cam.max_depth, "paths": st["paths"],
             "gpu_seconds": dt, "gpu_Msamples_per_s": st["paths"] / dt / 1e6, "gpu_Msegments_per_s": st["segments"] / dt / 1e6,
             "segments_per_path": st["segments"] / st["paths"], "light_probes_per_path": st["pdf_probes"] / st["paths"],
             "mean_radiance": float(accum.mean()) / n}
        scene.close()
        if not args.no_cpu:
            import helpers as H
            go = H.oracle_graph()
            wo, lo, camo = rt.scenes.build(name, go, prefix="orc_", lib=H.oracle_lib(), **kw)
            t0 = time.perf_counter()
            img, cnt = H.oracle_render(go, wo, lo, camo, sample_begin=0, sample_end=cpu_samples, threads=threads)
            dc = time.perf_counter() - t0
            r.update({"cpu_cores": threads, "cpu_samples_per_pixel_timed": cpu_samples, "cpu_seconds": dc,
                      "cpu_Msamples_per_s": cnt["paths"] / dc / 1e6, "cpu_Msegments_per_s": cnt["segments"] / dc / 1e6,
                      "cpu_segments_per_path": cnt["segments"] / cnt["paths"],
                      "cpu_mean_radiance": float(img.mean()) / cpu_samples,
                      "gpu_over_cpu": (st["paths"] / dt) / (cnt["paths"] / dc)})
        print(json.dumps(r), flush=True)
        results.append(r)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(results, f, indent=1)
    ctx.close()


if __name__ == "__main__":
    main()

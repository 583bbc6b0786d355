#!/usr/bin/env python3
"""One configuration rendered twice (warm-up, then the measured call), for a kernel trace of a single render call:

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_quick -o t -- python3 tools/trace_config.py quick
    python3 tools/trace_timeline.py gpurun_out/trace_quick

Prints the wall time of the measured call and the statistics of the context."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("name")
    ap.add_argument("--samples", type=int, default=0, help="sample indices to render (0 = the whole job)")
    ap.add_argument("--prims", type=int, default=1_000_000)
    ap.add_argument("--repeat", type=int, default=1)
    args = ap.parse_args()
    import raytracing_rust_amd as rt
    import hiprt
    ctx = rt.Context(0)
    g = rt.Graph()
    g.use_context(ctx)
    kw = {"n_prims": args.prims} if args.name == "stress" else {}
    world, lights, cam = rt.scenes.build(args.name, g, **kw)
    scene = rt.Scene.from_graph(ctx, g, world, lights)
    n = args.samples or cam.sqrt_spp ** 2
    acc = hiprt.DeviceArray(cam.image_width * cam.image_height * 3)
    out = []
    for k in range(1 + args.repeat):
        acc.zero()
        hiprt.synchronize()
        ctx.reset_stats()
        t0 = time.perf_counter()
        scene.render_device(cam, rt.render_opts(cam, sample_begin=0, sample_end=n, seed=rt.scenes.PATH_SEED + k), acc.ptr)
        hiprt.synchronize()
        dt = time.perf_counter() - t0
        st = ctx.stats()
        out.append({"call": k, "seconds": dt, "Msamples_per_s": st["paths"] / dt / 1e6, "Msegments_per_s": st["segments"] / dt / 1e6,
                    "paths": st["paths"], "segments": st["segments"], "iterations": st["iterations"]})
    print(json.dumps({"config": args.name, "samples": n, "calls": out}))
    scene.close()
    ctx.close()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Headline benchmark: Msamples/s on the reference's `final` scene (800x800, 4900-sample stratification, depth 75).

    python bench.py --gpus N --steps K --warmup W

A step is one pass of the hot path over one batch: ONE render call of the whole configuration the metric is quoted on,
every pixel of the 800x800 frame traced for all 70 x 70 = 4900 stratified sample indices (3.136 G camera paths;
`--rows R` renders R of the 70 rows per step instead).  Every (rank, step) gets its own path seed (weak scaling:
per-GPU work is fixed), all ranks' accumulators are summed on rank 0 with one reduce (RCCL) inside the timed region,
and rank 0 copies the sum to the host.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

WIDTH, SPP, DEPTH = 800, 5000, 75          # src/main.rs:22-29
# Algorithmic bytes per ray segment.  World-program traversal kernel (k_wf_traverse<true,false,1,true>, the kernel with the
# largest share of GPU time): reads the ray (origin, direction, time: 7 doubles), the path's RNG address because the
# scene has media (pixel, sample, bounce: 3 u32) and the hit the head-section launch left (t f64, kind_idx u32, ctx
# u32); writes the hit: 56 + 12 + 16 + 16 = 100 B (84 B when the head section is not walked separately).  A whole
# bounce adds the head-section launch (ray 56 in, hit 16 out), the deferred-section launch (4 + 56 + 16 in, 16 out
# for one ray in nine), the shading kernel's reads (16 B for its sort, then ray 56 + throughput 24 + ids 12 + hit 16 =
# 108 B unless the roulette ended the path) and its survivor record (80 + 12 = 92 B) or pixel atomics (24 B): about 390 B.
TRAVERSE_BYTES = 100
TRAVERSE_BYTES_UNSPLIT = 84
BOUNCE_BYTES = 390
# HBM-side traffic of that kernel per segment, from rocprofv3 PMC passes over this command with --rows 10 --steps 1
# --warmup 0 (1.0699e9 segments; profiles/r1_pmc_summary.txt): FETCH_SIZE 42.95e9 B doubled (the gfx950 half-count
# correction; it is calibrated for 16-byte loads and these are 8- and 4-byte ones, so an upper bound; the algorithmic
# reads are 52.3e9 B) + WRITE_SIZE 59.07e9 B (hit records 17.1e9 B, queue records of the new paths 41.2e9 B).
TRAVERSE_PMC_TRAFFIC_PER_SEGMENT = (2 * 42.95e9 + 59.07e9) / 1.0699e9
# The same for the shading kernel: FETCH_SIZE 135.8e9 B doubled + WRITE_SIZE 68.89e9 B (algorithmic: about 113e9 read,
# 57.2e9 of survivor records and dead entries + up to 10.8e9 of pixel atomics written).  The reads exceed the algorithmic bytes because
# the kernel shades each chunk of 4096 queue entries in sorted order: a 128-byte line of a queue array holds 16
# entries that are shaded by different waves at different times, and the chunks in flight (416 KB each, two per CU)
# outgrow the 4 MB L2 of an XCD, so lines come back from the Infinity Cache, which these counters count.
SHADE_PMC_TRAFFIC_PER_SEGMENT = (2 * 135.8e9 + 68.89e9) / 1.0699e9
HBM_PEAK_GBS = 8000.0


def cpu_baseline(threads, rows=1, samples=8):
    """The oracle (a C++ restatement of the reference, test infrastructure) on the host cores, bounded sample of the
    same workload: the full 800x800 frame, `samples` sample indices of the 4900, depth 75."""
    import raytracing_rust_amd as rt
    import helpers as H
    g = H.oracle_graph()
    w, l, cam = rt.scenes.build("final", g, width=WIDTH, spp=SPP, depth=DEPTH, prefix="orc_", lib=H.oracle_lib())
    t0 = time.perf_counter()
    _, cnt = H.oracle_render(g, w, l, cam, sample_begin=0, sample_end=samples, threads=threads)
    dt = time.perf_counter() - t0
    return {"value": cnt["paths"] / dt / 1e6, "unit": "Msamples/s", "cores": threads, "kind": "port",
            "sample": f"final scene 800x800, sample indices [0,{samples}) of 4900, depth 75: {cnt['paths']} paths in {dt:.1f} s",
            "Msegments_per_s": cnt["segments"] / dt / 1e6}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--rows", type=int, default=70, help="rows of the 70x70 stratification per step (a divisor of 70)")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pipeline", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-samples", type=int, default=256)
    args = ap.parse_args()

    import torch
    import raytracing_rust_amd as rt
    from raytracing_rust_amd import sharding

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world_size != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world_size}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the backend has no CPU path")
    # RTX_BENCH_REHEARSAL=1: every rank on GPU 0 and gloo collectives on host copies, to rehearse the multi-rank
    # control flow on a one-GPU box.  Not a measurement mode.
    rehearsal = os.environ.get("RTX_BENCH_REHEARSAL") == "1"
    device = 0 if rehearsal else local_rank
    torch.cuda.set_device(device)
    dist = None
    if world_size > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
    coll = "cpu" if rehearsal else "cuda"

    ctx = rt.Context(device)
    g = rt.Graph()
    world, lights, cam = rt.scenes.build("final", g, width=WIDTH, spp=SPP, depth=DEPTH)
    scene = rt.Scene.from_graph(ctx, g, world, lights)
    rows = cam.sqrt_spp                      # 70 sample indices per row, 70 rows
    npix = cam.image_width * cam.image_height
    accum = torch.zeros(npix * 3, dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def step(k):
        # batch unique per (rank, step); past the 70 rows of the stratification, go on with another seed
        b, e, seed = sharding.sample_batch(rank, k, args.steps + args.warmup, rows, rt.scenes.PATH_SEED, args.rows)
        opts = rt.render_opts(cam, sample_begin=b, sample_end=e, seed=seed, pipeline=args.pipeline)
        scene.render_device(cam, opts, accum, stream)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k)
    fence()
    ctx.reset_stats()
    accum.zero_()
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k)
    if rehearsal and dist is not None:
        host = accum.cpu()
        sharding.reduce_sum(host, dist, dst=0)
    else:
        sharding.reduce_sum(accum, dist, dst=0)
        if rank == 0:
            host = accum.cpu()
    fence()
    dt = time.perf_counter() - t0
    st = ctx.stats()
    if dist is not None:
        t = torch.tensor([dt, float(st["segments"]), float(st["paths"])], dtype=torch.float64, device=coll)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dt, segments, paths = float(tmax[0]), float(t[1]), float(t[2])
    else:
        segments, paths = float(st["segments"]), float(st["paths"])

    if rank == 0:
        assert bool(torch.isfinite(host).all()), "non-finite radiance in the accumulator"
        assert paths == npix * rows * args.rows * args.steps * world_size
        # Device-side timing of every traversal / shading launch (the library takes the wall clock of the first block in
        # and the last block out), rank 0.  The roofline entry is for whichever of the two took more GPU time.
        fused = st["traverse_launches"] == 0
        # Shading kernel, per queue entry: the sort reads the hit kind and the path's coordinates (4 + 12 B); an entry
        # the roulette does not end there is read in full (ray 56 + throughput 24 + ids 12 + hit 16 = 108 B) and then
        # either written back as a survivor (92 B) or ends with three pixel atomics (24 B).  A path ends exactly once.
        p_survive = 1.0 - paths / max(1.0, segments)
        p_drop = st.get("roulette_drops", 0) / max(1.0, st["segments"])
        shade_bytes = 16 + 108 * (1.0 - p_drop) + 92 * p_survive + 24 * max(0.0, 1.0 - p_survive - p_drop)
        cands = [("k_render_fused", st["render_ms"], st["launches"], BOUNCE_BYTES, None)] if fused else [
            # a new camera path (one segment in paths / segments) is not read at all: the world-program launch makes its
            # camera ray, writes its queue record (92 B) and its hit (16 B) and walks it from the start of the program
            ("k_wf_traverse<true,false,1,true>", st["traverse_ms"], st["traverse_launches"],
             TRAVERSE_BYTES + (108.0 - TRAVERSE_BYTES) * paths / max(1.0, segments) if st["head_launches"] else TRAVERSE_BYTES_UNSPLIT,
             TRAVERSE_PMC_TRAFFIC_PER_SEGMENT),
            ("k_wf_shade", st["shade_ms"], st["shade_launches"], shade_bytes, SHADE_PMC_TRAFFIC_PER_SEGMENT)]
        rl = []
        for name, ms, n_launch, seg_bytes, pmc in cands:
            kernel_ms = ms / max(1, n_launch)
            seg_per_launch = st["segments"] / max(1, n_launch)
            achieved = seg_per_launch * seg_bytes / (kernel_ms * 1e-3) / 1e9
            rl.append({"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                       "traffic": None if pmc is None else seg_per_launch * pmc,
                       "traffic_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, profiles/r1_pmc_summary.txt",
                       "kernel": name, "kernel_ms": kernel_ms, "launches": n_launch, "segments_per_launch": seg_per_launch,
                       "algorithmic_bytes_per_segment": seg_bytes, "algorithmic_bytes_per_launch": seg_per_launch * seg_bytes,
                       "gpu_ms": ms})
        rl.sort(key=lambda r: -r["gpu_ms"])
        roofline = dict(rl[0])
        roofline.update({
            "kernel_ms_source": "wall clock of the first block in / last block out of every launch, taken on the device "
                                "(a launch shares the GPU with the launches of the other pipelines)",
            "whole_bounce": {"algorithmic_bytes_per_segment": BOUNCE_BYTES,
                             "achieved": segments / world_size / dt * BOUNCE_BYTES / 1e9, "unit": "GB/s",
                             "frac": segments / world_size / dt * BOUNCE_BYTES / 1e9 / HBM_PEAK_GBS},
            "other_kernels": rl[1:],
            "note": "f64 VALU/latency-bound code; the HBM streams of rays, hits and path state are the only bytes it must move; "
                    "four pipelines run side by side, so a launch does not have the GPU to itself"})
        out = {
            "metric": "Msamples/s (node) final scene 800x800", "value": paths / dt / 1e6, "unit": "Msamples/s",
            "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"final scene (final_scene.rs) 800x800, depth 75, step = one render call of {args.rows} of the "
                                   f"70 rows of the 4900-sample stratification = {npix * rows * args.rows / 1e6:.1f}M camera "
                                   "paths per GPU per step",
                       "scene_seed": hex(rt.scenes.SCENE_SEED), "sharding": "sample rows per rank, one reduce(sum) of the f64 accumulator",
                       "pipelines_per_gpu": int(os.environ.get("RTX_WF_LANES", "4")),
                       "paths_in_flight_per_pipeline": int(os.environ.get("RTX_WF_CAPACITY", str(1 << 27)))},
            "Msegments_per_s": segments / dt / 1e6, "segments_per_path": segments / paths,
            "roofline": roofline,
        }
        if world_size == 1 and not args.no_cpu_baseline:
            threads = min(16, os.cpu_count() or 1)
            out["cpu_baseline"] = cpu_baseline(threads, samples=args.cpu_samples)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    scene.close()
    ctx.close()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Headline benchmark: Msamples/s on the reference's `final` scene (800x800, 4900-sample stratification, depth 75).

    python bench.py --gpus N --steps K --warmup W

A step is one pass of the hot path over one batch: ONE render call of the whole configuration the metric is quoted on,
every pixel of the 800x800 frame traced for all 70 x 70 = 4900 stratified sample indices (3.136 G camera paths;
`--rows R` renders R of the 70 rows per step instead).

With N > 1 the command starts its own ranks (one process per GPU through torch.distributed.run; the parent process
never touches a GPU) unless it already runs under a launcher (WORLD_SIZE set).  Modes, each timed over K steps between
barriers, maximum over ranks:
  strong  (the headline `value` at every N): ONE frame per step, its sample indices split evenly between the ranks
          (by index: 4900 over 8 ranks = 613 or 612 each), one reduce(sum) of the f64 accumulators (RCCL) and the copy
          to the host after every frame, inside the timed region: the 800x800x5000spp image on N GPUs.
  weak    (`weak_scaling`, N > 1 only): every (rank, step) renders a whole frame with its own path seed; one reduce
          and copy to the host at the end of the timed region.
Rank 0 then measures the roofline block (a short single-pipeline render, so that kernels run one at a time and their
durations are exclusive) and, at N = 1, the CPU baseline.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

WIDTH, SPP, DEPTH = 800, 5000, 75          # src/main.rs:22-29
HBM_PEAK_GBS = 8000.0
# Algorithmic bytes per ray segment (one world.hit of ray_color, camera.rs:179).
#   a-priori figures for an f64 wavefront layout: traversal kernel 80 B (ray origin, direction, time, tmax = 64 in;
#   hit t, primitive, flags = 16 out); whole bounce 328 B (traversal 80 + shading reads them back 80 + path state 40 in,
#   next ray 64 + state 40 out, 24 of radiance).
#   what this build's kernels move: the world-program traversal reads ray 56 + RNG address 12 (the scene has media) +
#   the hit the head-section launch left 16 and writes the hit 16 = 100 B for a surviving path; for a NEW camera path
#   it reads nothing and writes the queue record 92 + hit 16 = 108 B.  Head-section stream: ray 56 in, hit 16 out.
#   Deferred-section stream: index 4 + ray 56 + hit 16 in, hit 16 out, per ray that entered the section.  Shading: 16 B
#   per entry for the sort, 108 B in for an entry the roulette did not end, then a survivor record 92 B or 24 B of
#   pixel atomics.  Whole bounce: about 390 B.
APRIORI_TRAVERSE_BYTES, APRIORI_BOUNCE_BYTES = 80, 328
OWN_TRAVERSE_BYTES, OWN_NEW_PATH_BYTES, OWN_BOUNCE_BYTES = 100, 108, 390
# Counter-derived constants of the profiled build (rocprofv3 --pmc passes over `RTX_WF_LANES=1 python bench.py --rows 10
# --steps 1 --warmup 0 --roofline-rows 0 --no-cpu-baseline`, tools/prof_gpu.sh; profiles/README.md names the commit).  Per ray segment of the whole
# pipeline: VALU wave instructions, and per kernel the HBM-side bytes (FETCH_SIZE doubled as the gfx950 correction for
# wide loads prescribes - an upper bound here, the loads are 4 and 8 bytes per lane - plus WRITE_SIZE) and the lanes
# active per VALU instruction.
PMC = {
    "profile": "profiles/r3_pmc_summary.txt (one step of 10 rows, 1.06993e9 segments, one pipeline; tools/pmc_derive.py)",
    "valu_wave_instructions_per_segment": 69.8,
    # (2 x FETCH_SIZE + WRITE_SIZE) / segments; FETCH_SIZE as reported would give 79.4, 190.1, 40.9 and 9.9
    "traffic_bytes_per_segment": {"k_wf_traverse": 103.8, "k_wf_shade": 314.0, "k_wf_stream_head": 58.0, "k_wf_stream_deferred": 15.8},
    "lanes_active": {"k_wf_traverse": 57.0, "k_wf_stream_head": 42.2, "k_wf_stream_deferred": 34.8, "k_wf_shade": 41.0},
}
SIMDS, CLOCK_GHZ, F64_CYCLES_PER_WAVE_INSTRUCTION = 256 * 4, 2.4, 4   # vector f64: 16 lanes per clock per SIMD


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(samples=8):
    """The oracle (a C++ restatement of the reference, test infrastructure) on the host cores this job may use, on a
    bounded sample of the same workload: the full 800x800 frame, `samples` sample indices of the 4900, depth 75."""
    import raytracing_rust_amd as rt
    import helpers as H
    nproc = os.cpu_count() or 1
    granted = nproc
    try:   # a GPU box shows the whole machine's CPUs; the job's share is what its launcher exports
        granted = min(nproc, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    if os.environ.get("OMP_NUM_THREADS", "").isdigit():
        granted = max(1, min(granted, int(os.environ["OMP_NUM_THREADS"])))
    g = H.oracle_graph()
    w, l, cam = rt.scenes.build("final", g, width=WIDTH, spp=SPP, depth=DEPTH, prefix="orc_", lib=H.oracle_lib())
    t0 = time.perf_counter()
    _, cnt = H.oracle_render(g, w, l, cam, sample_begin=0, sample_end=samples, threads=granted)
    dt = time.perf_counter() - t0
    return {"value": cnt["paths"] / dt / 1e6, "unit": "Msamples/s", "cores": granted, "kind": "port",
            "sample": f"final scene 800x800, sample indices [0,{samples}) of 4900, depth 75: {cnt['paths']} paths in {dt:.1f} s",
            "nproc": nproc, "cpu_model": cpu_model(),
            "cores_note": "every core this job may use (affinity mask, capped by the launcher's OMP_NUM_THREADS)",
            "Msegments_per_s": cnt["segments"] / dt / 1e6}


def spawn_ranks(args):
    """--gpus N without a launcher: start the N ranks as a child process tree and pass its output and exit code on.
    This process makes no GPU call before or after."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def roofline_block(rt, ctx, scene, cam, accum, stream, rows, wall_seg_per_s, torch):
    """Exclusive kernel durations: `rows` rows of the stratification through ONE pipeline, so that no two kernels
    share the GPU.  World-program traversal and shading are timed on the device (wall clock of the first block in and
    the last block out of every launch), the two section streams with HIP events on the stream they are launched on."""
    prev = os.environ.get("RTX_WF_LANES")
    os.environ["RTX_WF_LANES"] = "1"
    try:
        torch.cuda.synchronize()
        ctx.reset_stats()
        accum.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        scene.render_device(cam, rt.render_opts(cam, sample_begin=0, sample_end=rows * cam.sqrt_spp, seed=rt.scenes.PATH_SEED + 99), accum, stream)
        torch.cuda.synchronize()
        wall_ms = (time.perf_counter() - t0) * 1e3
        st = ctx.stats()
    finally:
        if prev is None:
            del os.environ["RTX_WF_LANES"]
        else:
            os.environ["RTX_WF_LANES"] = prev
    seg, paths, it = float(st["segments"]), float(st["paths"]), max(1, st["iterations"])
    if st["traverse_launches"] == 0:   # the fused pipeline was forced
        ms = st["render_ms"]
        ach = seg * APRIORI_BOUNCE_BYTES / (ms * 1e-3) / 1e9
        return {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                "kernel": "k_render_fused", "kernel_ms": ms / max(1, st["launches"]), "algorithmic_bytes_per_segment": APRIORI_BOUNCE_BYTES}
    new_frac = paths / seg
    own_trav = OWN_TRAVERSE_BYTES * (1.0 - new_frac) + OWN_NEW_PATH_BYTES * new_frac
    p_drop = st["roulette_drops"] / seg
    p_survive = 1.0 - paths / seg
    own_shade = 16 + 108 * (1.0 - p_drop) + 92 * p_survive + 24 * max(0.0, 1.0 - p_survive - p_drop)
    # deferred rays are not counted by the library; one ray in nine enters the section in this scene (profiles/README.md)
    kernels = [
        ("k_wf_traverse<true,false,1,true,true>", "world program + camera rays", st["traverse_ms"], APRIORI_TRAVERSE_BYTES, own_trav, "k_wf_traverse"),
        ("k_wf_shade", "shading + compaction", st["shade_ms"], APRIORI_BOUNCE_BYTES - APRIORI_TRAVERSE_BYTES, own_shade, "k_wf_shade"),
        ("k_wf_stream<true,false,0>", "head section (ground BVH), survivors", st["head_ms"], APRIORI_TRAVERSE_BYTES, 72.0 * (1.0 - new_frac), "k_wf_stream_head"),
        ("k_wf_stream<true,true,1>", "deferred section (sphere cluster)", st["tail_ms"], APRIORI_TRAVERSE_BYTES / 9.0, 92.0 / 9.0, "k_wf_stream_deferred"),
    ]
    total_ms = sum(k[2] for k in kernels)
    trav_ms = max(1e-9, sum(k[2] for k in kernels if k[5] != "k_wf_shade"))
    out = []
    for name, what, ms, survey_b, own_b, key in kernels:
        if ms <= 0.0:
            continue
        ach = seg * survey_b / (ms * 1e-3) / 1e9
        traffic = PMC["traffic_bytes_per_segment"].get(key)
        out.append({"kernel": name, "what": what, "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "achieved": ach, "frac": ach / HBM_PEAK_GBS,
                    "algorithmic_bytes_per_segment": survey_b, "algorithmic_bytes_per_launch": survey_b * seg / it,
                    "achieved_own_bytes": seg * own_b / (ms * 1e-3) / 1e9, "frac_own_bytes": seg * own_b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "own_bytes_per_segment": own_b,
                    "kernel_ms": ms / it, "launches": it, "segments_per_launch": seg / it, "gpu_ms": ms,
                    "share_of_kernel_time": ms / total_ms,
                    "traffic": None if traffic is None else traffic * seg / it,
                    "lanes_active_per_valu_instruction": PMC["lanes_active"].get(key)})
    out.sort(key=lambda r: -r["gpu_ms"])
    rl = dict(out[0])
    seg_s = seg / (wall_ms * 1e-3)
    rl.update({
        "kernel_ms_source": f"single-pipeline pass (RTX_WF_LANES=1) over {rows} of the 70 rows: kernels run one at a time, so durations "
                            "are exclusive; device wall clock per launch for traversal and shading, HIP events on the launch "
                            "stream for the section streams; averages are over every launch, empty ones included, as a profiler counts",
        "traffic_source": f"rocprofv3 --pmc FETCH_SIZE (doubled) + WRITE_SIZE passes, {PMC['profile']}",
        "pass": {"rows": rows, "wall_ms": wall_ms, "sum_kernel_ms": total_ms, "segments": seg, "paths": paths,
                 "Msegments_per_s": seg_s / 1e6, "Msamples_per_s": paths / (wall_ms * 1e-3) / 1e6},
        "other_kernels": out[1:],
        # one world.hit of a segment is spread over three kernels (head section, world program, deferred section):
        # the same 80 B per segment against the sum of their durations
        "traversal_total": {"kernels": [k[0] for k in kernels if k[5] != "k_wf_shade" and k[2] > 0.0],
                            "algorithmic_bytes_per_segment": APRIORI_TRAVERSE_BYTES, "gpu_ms": trav_ms,
                            "achieved": seg * APRIORI_TRAVERSE_BYTES / (trav_ms * 1e-3) / 1e9,
                            "frac": seg * APRIORI_TRAVERSE_BYTES / (trav_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "unit": "GB/s",
                            "Msegments_per_s": seg / (trav_ms * 1e-3) / 1e6},
        "whole_bounce": {"note": "all kernels of a bounce over the WALL time of the timed region (all pipelines)",
                         "algorithmic_bytes_per_segment": APRIORI_BOUNCE_BYTES,
                         "achieved": wall_seg_per_s * APRIORI_BOUNCE_BYTES / 1e9, "frac": wall_seg_per_s * APRIORI_BOUNCE_BYTES / 1e9 / HBM_PEAK_GBS,
                         "own_bytes_per_segment": OWN_BOUNCE_BYTES,
                         "achieved_own_bytes": wall_seg_per_s * OWN_BOUNCE_BYTES / 1e9, "frac_own_bytes": wall_seg_per_s * OWN_BOUNCE_BYTES / 1e9 / HBM_PEAK_GBS,
                         "unit": "GB/s"},
        # what actually bounds these kernels: f64 VALU issue with partly filled waves
        "valu": {"bound": "f64 VALU issue", "wave_instructions_per_segment": PMC["valu_wave_instructions_per_segment"],
                 "cycles_per_wave_instruction": F64_CYCLES_PER_WAVE_INSTRUCTION, "simds": SIMDS, "clock_ghz": CLOCK_GHZ,
                 "issue_utilisation": wall_seg_per_s * PMC["valu_wave_instructions_per_segment"] * F64_CYCLES_PER_WAVE_INSTRUCTION / (SIMDS * CLOCK_GHZ * 1e9),
                 "lanes_active_of_64": PMC["lanes_active"], "source": PMC["profile"],
                 "note": "share of the issue slots of 1024 SIMDs at 2.4 GHz over the timed region, every VALU instruction "
                         "priced as vector f64 (4 cycles per wave); the count per segment is a counter measurement of the profiled build"},
        "note": "f64 VALU- and latency-bound code: the scene is LDS/L2-resident, the HBM streams of rays, hits and path state are "
                "the only bytes it must move"})
    return rl


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--rows", type=int, default=70, help="rows of the 70x70 stratification per step (a divisor of 70)")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pipeline", type=int, default=0)
    ap.add_argument("--scaling", choices=["both", "weak", "strong"], default="both", help="modes to time when --gpus > 1")
    ap.add_argument("--roofline-rows", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-samples", type=int, default=256)
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))

    # The benchmark has the GPU to itself: queues of 128 M paths per pipeline (118 GB of the 288 GB) instead of the
    # library's default of 64 M (59 GB), worth about 2 %.  Reported in config.paths_in_flight_per_pipeline.
    os.environ.setdefault("RTX_WF_CAPACITY", str(1 << 27))
    import torch   # before the package: in a process that holds both, PyTorch's HIP runtime has to be loaded first
    import raytracing_rust_amd as rt
    from raytracing_rust_amd import sharding

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world_size != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world_size}")
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
    host = None

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def reduce_to_host():
        if rehearsal and dist is not None:
            h = accum.cpu()
            sharding.reduce_sum(h, dist, dst=0)
            return h
        sharding.reduce_sum(accum, dist, dst=0)
        return accum.cpu() if rank == 0 else None

    def weak_step(k):
        # batch unique per (rank, step); past the 70 rows of the stratification, go on with another seed
        b, e, seed = sharding.sample_batch(rank, k, args.steps + args.warmup, rows, rt.scenes.PATH_SEED, args.rows)
        scene.render_device(cam, rt.render_opts(cam, sample_begin=b, sample_end=e, seed=seed, pipeline=args.pipeline), accum, stream)

    def strong_step(k):
        # frame k: its sample indices split between the ranks, then one reduce and the copy to the host
        first, end = sharding.rank_samples(rank, world_size, args.rows * rows)
        accum.zero_()
        if end > first:
            scene.render_device(cam, rt.render_opts(cam, sample_begin=first, sample_end=end,
                                                    seed=rt.scenes.PATH_SEED + 1000 + k, pipeline=args.pipeline), accum, stream)
        return reduce_to_host()

    def timed(step_fn, finish):
        for k in range(args.warmup):
            step_fn(k)
        fence()
        ctx.reset_stats()
        accum.zero_()
        fence()
        t0 = time.perf_counter()
        last = None
        for k in range(args.steps):
            last = step_fn(args.warmup + k)
        if finish is not None:
            last = finish()
        fence()
        dt = time.perf_counter() - t0
        st = ctx.stats()
        seg, paths = float(st["segments"]), float(st["paths"])
        if dist is not None:
            t = torch.tensor([dt, seg, paths], dtype=torch.float64, device=coll)
            tmax = t.clone()
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            dt, seg, paths = float(tmax[0]), float(t[1]), float(t[2])
        return dt, seg, paths, last

    results = {}
    if args.scaling in ("both", "strong") or world_size == 1:
        results["strong"] = timed(strong_step, None)
    if world_size > 1 and args.scaling in ("both", "weak"):
        results["weak"] = timed(weak_step, reduce_to_host)
    mode = "strong" if "strong" in results else "weak"
    dt, segments, paths, host = results[mode]

    if rank == 0:
        assert host is not None and bool(torch.isfinite(host).all()), "non-finite radiance in the accumulator"
        per_step = npix * rows * args.rows
        assert paths == per_step * args.steps * (world_size if mode == "weak" else 1), (paths, per_step)
        out = {
            "metric": "Msamples/s (node) final scene 800x800", "value": paths / dt / 1e6, "unit": "Msamples/s",
            "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": mode, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"final scene (final_scene.rs) 800x800, depth 75, step = one render call of {args.rows} of the "
                                   f"70 rows of the 4900-sample stratification = {per_step / 1e6:.1f}M camera paths",
                       "strong": "every step is ONE such batch (one frame), its sample indices split evenly between the ranks; "
                                 "reduce(sum) of the f64 accumulators and copy to the host after every step, inside the timed region",
                       "weak": "every (rank, step) renders that batch with its own path seed; one reduce(sum) and the copy to "
                               "the host at the end of the timed region",
                       "scene_seed": hex(rt.scenes.SCENE_SEED),
                       "pipelines_per_gpu": int(os.environ.get("RTX_WF_LANES", "4")),
                       "paths_in_flight_per_pipeline": int(os.environ.get("RTX_WF_CAPACITY", str(1 << 27)))},
            "Msegments_per_s": segments / dt / 1e6, "segments_per_path": segments / paths,
        }
        if "weak" in results and mode != "weak":
            wdt, wseg, wpaths, whost = results["weak"]
            assert whost is not None and bool(torch.isfinite(whost).all()) and wpaths == per_step * args.steps * world_size
            out["weak_scaling"] = {"value": wpaths / wdt / 1e6, "unit": "Msamples/s", "ms_per_step": wdt / args.steps * 1e3,
                                   "steps": args.steps, "Msegments_per_s": wseg / wdt / 1e6,
                                   "what": "a whole 800x800 frame per rank and step, one reduce + copy to the host at the end"}
    # rank 0 alone: exclusive kernel durations for the roofline block (the other ranks wait at the barrier below)
    if rank == 0:
        if args.roofline_rows > 0:   # 0: a profiler run that wants the timed region alone
            out["roofline"] = roofline_block(rt, ctx, scene, cam, accum, stream, min(args.roofline_rows, rows),
                                             segments / world_size / dt, torch)
        if world_size == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(samples=args.cpu_samples)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    scene.close()
    ctx.close()


if __name__ == "__main__":
    main()

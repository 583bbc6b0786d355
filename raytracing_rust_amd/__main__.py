"""Entry points of the reference's binary (src/main.rs:8-45): `cornell`, `final`, `quick`, plus `smoke`.

    python -m raytracing_rust_amd cornell|final|quick|smoke [--width W] [--spp N] [--depth D] [--out FILE]

Builds the scene through the recording constructors, renders every sample of every pixel on the GPU, resolves to
RGB8 with the reference's gamma and spp scale (color.rs:16-52) and writes the PNG the reference would write.
"""
import argparse
import ctypes as C
import sys

from . import _capi, scenes
from .graph import Graph
from .render import Context, RtxError

OUTPUT = {"cornell": "cornell_box_glass.png", "final": "final_scene.png", "quick": "quick_test.png",
          "smoke": "cornell_smoke.png"}                                   # main.rs:18,27,37


def main(argv=None):
    ap = argparse.ArgumentParser(prog="raytracing_rust_amd", description=__doc__.splitlines()[0])
    ap.add_argument("scene", nargs="?", choices=sorted(OUTPUT))
    ap.add_argument("--width", type=int)
    ap.add_argument("--spp", type=int)
    ap.add_argument("--depth", type=int)
    ap.add_argument("--out")
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=scenes.PATH_SEED)
    ap.add_argument("--checkpoint", help="render in batches and keep the accumulator in this file; resumes if it exists")
    ap.add_argument("--rows-per-batch", type=int, default=5, help="rows of the stratification per batch (with --checkpoint)")
    args = ap.parse_args(argv)
    if args.scene is None:                                                # main.rs:39-44
        print("usage: python -m raytracing_rust_amd [cornell|final|quick|smoke]", file=sys.stderr)
        print("  cornell - Cornell box with a glass sphere", file=sys.stderr)
        print("  final   - the final scene", file=sys.stderr)
        print("  quick   - the final scene at 400x400, 100 spp", file=sys.stderr)
        return 2
    g = Graph()
    world, lights, cam = scenes.build(args.scene, g, width=args.width, spp=args.spp, depth=args.depth)
    ctx = Context(0)
    out = (args.out or OUTPUT[args.scene]).encode()
    secs = C.c_double(0.0)
    if args.checkpoint:
        import time
        t0 = time.perf_counter()

        def report(rows, total, seconds, _user):
            print(f"  rows {rows}/{total}  {seconds:7.1f} s", file=sys.stderr)
            return 0
        cb = _capi.PROGRESS_FN(report)
        st = _capi.lib.rtxh_camera_render_progressive(g.session, ctx.handle, C.byref(cam), world,
                                                      -1 if lights is None else lights, args.seed, args.rows_per_batch,
                                                      args.checkpoint.encode(), cb, None, out, None, None)
        secs.value = time.perf_counter() - t0
    else:
        st = _capi.lib.rtxh_camera_render(g.session, ctx.handle, C.byref(cam), world, -1 if lights is None else lights,
                                          args.seed, out, None, C.byref(secs))
    if st != _capi.RTX_OK:
        msg = _capi.lib.rtx_last_error(ctx.handle) or _capi.lib.rtxh_last_error(g.session)
        raise RtxError(st, msg.decode() if msg else "")
    n = cam.image_width * cam.image_height * cam.sqrt_spp ** 2
    print(f"{args.scene}: {cam.image_width}x{cam.image_height}, {cam.sqrt_spp ** 2} samples per pixel, depth {cam.max_depth}: "
          f"{secs.value:.2f} s, {n / secs.value / 1e6:.1f} Msamples/s -> {out.decode()}")
    ctx.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
